#!/bin/bash
# cloud iteration: media parity tests (optional), bench line of the cloud, kernel stats
#   tools/gpu_cloud.sh <tag> [test]
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; mkdir -p $O
if [ "$2" == "test" ]; then
timeout 1500 python -m pytest "tests/test_parity_holes.py::test_full_size_cloud" "tests/test_parity_holes.py::test_medium_pointwise_bit_exact" "tests/test_parity_holes.py::test_scheduling_is_result_neutral" tests/test_gpu_parity.py::test_media_frame_parity_statistical "tests/test_converged_parity.py::test_bomex_crop_converged_parity" "tests/test_converged_parity.py::test_media_converged_parity" -m gpu -x -q --timeout 900 2>&1 | tail -15
fi
for c in cloud; do
  timeout 900 python bench.py --config $c --no-cpu-baseline > $O/bench_$c.json 2> $O/bench_$c.err
  python - <<PY
import json
try:
    d = json.load(open("$O/bench_$c.json"))
    print("$c", d["value"], "Mrays/s", d["seconds_per_frame"], "s/frame cold", d["cold_frame_seconds"], d["roofline"]["kernel_seconds"])
    for e in d["rooflines"]: print("   ", e["kernel"], e["frac"], e["seconds"], e.get("units"))
except Exception as e:
    print("$c FAILED", e); print(open("$O/bench_$c.err").read()[-2500:])
PY
done
rocprofv3 --kernel-trace --stats -d $O/trace_cloud -- python3 bench.py --config cloud --no-cpu-baseline --steps 1 --warmup 1 > $O/trace_cloud.log 2>&1
python3 tools/rocpd_summary.py $O/trace_cloud/*/*_results.db > $O/trace_cloud.txt 2>&1
find $O -name "*_results.db" -delete
head -8 $O/trace_cloud.txt | cut -c1-160

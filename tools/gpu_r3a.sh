#!/bin/bash
# round-3 first GPU pass: new tests, then bench lines of cloud + cornell, kernel stats of the cloud
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r3a; mkdir -p $O
echo skip tests > $O/tests.txt
tail -25 $O/tests.txt
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
head -20 $O/trace_cloud.txt | cut -c1-160

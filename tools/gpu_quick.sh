#!/bin/bash
# Quick GPU iteration on the MI355X box: (optional) GPU tests, bench lines of the chosen configs, kernel-stats summaries.
#   tools/gpu_quick.sh <tag> "<configs>" [test] [prof "<configs>"]
# Outputs land in gpurun_out/<tag>/ (scratch; copy what is to be judged into profiles/).
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
TAG=$1; CONFIGS=$2; shift 2
O=gpurun_out/$TAG; mkdir -p $O
if [ "$1" == "test" ]; then shift; python -m pytest tests -m gpu -x -q --timeout 900 2>&1 | tail -6; fi
for c in $CONFIGS; do
  timeout 900 python bench.py --config $c --no-cpu-baseline > $O/bench_$c.json 2> $O/bench_$c.err
  python - <<PY
import json
try:
    d = json.load(open("$O/bench_$c.json"))
    print("$c", d["value"], "Mrays/s", d["seconds_to_256spp"], "s/256spp", d["roofline"]["kernel_seconds"])
except Exception as e:
    print("$c FAILED", e); print(open("$O/bench_$c.err").read()[-1500:])
PY
done
if [ "$1" == "prof" ]; then
  for c in $2; do
    steps=""; if [ "$c" == "cloud" ] || [ "$c" == "manylight" ]; then steps="--steps 1 --warmup 1"; fi
    rocprofv3 --kernel-trace --stats -d $O/trace_$c -- python3 bench.py --config $c --no-cpu-baseline $steps > $O/trace_$c.log 2>&1
    python3 tools/rocpd_summary.py $O/trace_$c/*/*_results.db > $O/trace_$c.txt 2>&1
    find $O/trace_$c -name "*_results.db" -delete
    head -14 $O/trace_$c.txt | cut -c1-150
  done
fi

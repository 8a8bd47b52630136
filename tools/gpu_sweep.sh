#!/bin/bash
# sweep an environment knob over the cloud bench:  tools/gpu_sweep.sh <tag> <VAR> "<values>" [config]
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; mkdir -p $O; VAR=$2; CFG=${4:-cloud}
for v in $3; do
  env $VAR=$v timeout 900 python bench.py --config $CFG --no-cpu-baseline --warmup 1 > $O/bench_${CFG}_$v.json 2> $O/bench_${CFG}_$v.err
  python - <<PY
import json
try:
    d = json.load(open("$O/bench_${CFG}_$v.json"))
    print("$VAR=$v", d["value"], "Mrays/s", d["seconds_per_frame"], "s/frame", d["roofline"]["kernel_seconds"])
except Exception as e:
    print("$VAR=$v FAILED", e); print(open("$O/bench_${CFG}_$v.err").read()[-1500:])
PY
done

#!/bin/bash
# sweep "VAR=v VAR2=v2" settings over one bench config:  tools/gpu_sweep2.sh <tag> <config> "<setting>" "<setting>" ...
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; mkdir -p $O; CFG=$2; shift 2
i=0
for setting in "$@"; do
  i=$((i+1))
  env $setting timeout 900 python bench.py --config $CFG --no-cpu-baseline --warmup 1 > $O/bench_$i.json 2> $O/bench_$i.err
  python - <<PY
import json
try:
    d = json.load(open("$O/bench_$i.json"))
    print("$setting |", d["value"], "Mrays/s", d["seconds_per_frame"], "s/frame", d["roofline"]["kernel_seconds"])
except Exception as e:
    print("$setting FAILED", e); print(open("$O/bench_$i.err").read()[-1500:])
PY
done

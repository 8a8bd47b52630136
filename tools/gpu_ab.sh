#!/bin/bash
# A/B two library builds over bench configs:  tools/gpu_ab.sh "<configs>" <libA> <libB> [repeats]
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
R=${4:-1}
for c in $1; do
for rep in $(seq 1 $R); do
for lib in $2 $3; do
  HK_LIB_PATH=$lib timeout 900 python bench.py --config $c --no-cpu-baseline --warmup 1 > /tmp/b.json 2> /tmp/b.err
  python - <<PY
import json
try:
    d = json.load(open("/tmp/b.json"))
    ks = d["roofline"]["kernel_seconds"]; n = d["steps"]
    print("$c $lib |", d["value"], "Mrays/s", d["seconds_per_frame"], "s/frame", {k: round(v / n, 5) for k, v in ks.items()})
except Exception as e:
    print("$c $lib FAILED", e); print(open("/tmp/b.err").read()[-1500:])
PY
done; done; done

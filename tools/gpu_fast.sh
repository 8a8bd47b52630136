#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for c in cornell sky manylight cloud; do
for lib in hikari.jl_amd/csrc/libhikari_mi355x.so hikari.jl_amd/csrc/libhikari_mi355x_fast.so; do
  HK_LIB_PATH=$lib timeout 900 python bench.py --config $c --no-cpu-baseline --warmup 1 > /tmp/b.json 2> /tmp/b.err
  python - <<PY
import json
try:
    d = json.load(open("/tmp/b.json"))
    print("$c $lib |", d["value"], "Mrays/s", d["seconds_per_frame"], "s/frame", d["roofline"]["kernel_seconds"])
except Exception as e:
    print("$c $lib FAILED", e); print(open("/tmp/b.err").read()[-1500:])
PY
done; done

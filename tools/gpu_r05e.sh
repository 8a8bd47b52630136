#!/bin/bash
# SQ counters of the many-light frame under both light-select kernels
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for pool in 1 0; do
  export HK_SELECT_POOL=$pool
  tools/gpu_sq.sh r05e_pool$pool manylight > gpurun_out/r05e_pool$pool.log 2>&1
  python3 - <<P
import json
d=json.load(open("gpurun_out/r05e_pool$pool/utilisation_manylight.json"))
for k in ("k_light_select","k_shade","k_trace","k_shadow"):
    print("pool=$pool", k, d.get(k))
P
done

#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for spp in 32 64; do
for w in 0 16 24 40 64 128; do
  HK_WAVES_PER_CU=$w timeout 600 python bench.py --no-cpu-baseline --spp $spp --steps 40 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print($spp, $w, d['ms_per_step'], d['value'])"
done; done | tee gpurun_out/spp_waves.txt

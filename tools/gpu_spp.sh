#!/bin/bash
# per-frame fixed cost: one GPU at 256 / 128 / 64 / 32 spp per frame (what a rank of 1 / 2 / 4 / 8 renders under strong scaling)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for spp in 256 128 64 32; do
  timeout 600 python bench.py --no-cpu-baseline --spp $spp --steps 40 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print($spp, d['ms_per_step'], d['value'])"
done | tee gpurun_out/spp_scaling.txt
# two ranks sharing the device, strong scaling: the reduce and rendezvous cost on top
HK_BENCH_SINGLE_DEVICE=1 timeout 600 python bench.py --no-cpu-baseline --gpus 2 --steps 20 2>/dev/null | tail -1 | cut -c1-400 | tee -a gpurun_out/spp_scaling.txt

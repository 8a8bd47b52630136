#!/bin/bash
# tools/gpu_spp.sh [tag]   (default r06) — a rank's share of the 8-GPU configs on one GPU (cloud 256 / 128 / 64 / 32 spp, many-light 512 / 256 / 128 / 64): ms per frame, Mrays/s
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for spec in "cloud 256 3" "cloud 128 4" "cloud 64 6" "cloud 32 10" "manylight 512 2" "manylight 256 3" "manylight 128 5" "manylight 64 8"; do
  set -- $spec
  timeout 900 python bench.py --config $1 --no-cpu-baseline --progressive 0 --spp $2 --steps $3 --warmup 1 --detail-file /tmp/spp5.json > /dev/null 2>/tmp/spp5.err || tail -2 /tmp/spp5.err
  python3 -c "
import json
d=json.load(open('/tmp/spp5.json')); print('$1', $2, d['ms_per_step'], d['value'], {k: round(v/d['steps']*1e3,2) for k,v in d['roofline']['kernel_seconds'].items()})"
done | tee gpurun_out/${1:-r06}_spp_scaling_raw.txt

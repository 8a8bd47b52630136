#!/bin/bash
# many-light frame under the variants of the pooled light-select kernel (A/B on one box, two rounds), bit-identity first
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_parity_holes.py::test_light_preselection_is_result_neutral tests/test_gpu_parity.py::test_light_bvh_parity "tests/test_gpu_parity.py::test_frame_parity" -m gpu -q --timeout 800 2>&1 | tail -3
for rep in 1 2; do
for spec in - scratch/lib_sel4.so scratch/lib_selb256.so scratch/lib_selb1024.so scratch/lib_selnotop.so HK_SELECT_POOL=0; do
  envs=""; case $spec in -) ;; *=*) envs="$spec";; *) envs="HK_LIB_PATH=$spec";; esac
  env $envs timeout 600 python bench.py --config manylight --no-cpu-baseline --progressive 0 --warmup 1 --detail-file /tmp/d.json > /dev/null 2>&1
  python3 -c "
import json
d=json.load(open('/tmp/d.json')); print('$spec', d['seconds_per_frame'], d['value'], d['roofline']['kernel_seconds'])"
done; done

#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests/test_gpu_parity.py tests/test_golden.py tests/test_layered_materials.py tests/test_parity_holes.py tests/test_environment_light.py -m gpu -x -q --timeout 1200 2>&1 | tail -4
for c in cornell sky manylight cloud; do tools/gpu_sweep2.sh r3y $c "HK_X=1"; done

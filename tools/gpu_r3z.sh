#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests/test_gpu_parity.py "tests/test_parity_holes.py" -m gpu -x -q --timeout 1200 2>&1 | tail -4
tools/gpu_sweep2.sh r3z manylight "HK_PRESELECT=1" "HK_PRESELECT=0"

#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests/test_gpu_parity.py tests/test_golden.py "tests/test_parity_holes.py::test_f64_film_frame" -m gpu -x -q --timeout 1200 2>&1 | tail -4
tools/gpu_sweep2.sh r3s cornell "HK_X=1"
tools/gpu_sweep2.sh r3s sky "HK_X=1"

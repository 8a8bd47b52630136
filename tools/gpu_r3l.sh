#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest "tests/test_parity_holes.py::test_full_size_cloud" "tests/test_parity_holes.py::test_scheduling_is_result_neutral" tests/test_gpu_parity.py::test_media_frame_parity_statistical "tests/test_converged_parity.py::test_bomex_crop_converged_parity" "tests/test_converged_parity.py::test_media_converged_parity" -m gpu -x -q --timeout 900 2>&1 | tail -15
tools/gpu_sweep2.sh r3l cloud "HK_WALK_SPLIT=1" "HK_WALK_SPLIT=0"

#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 3000 python -m pytest tests/test_multi_gpu.py tests/test_parity_holes.py tests/test_control_flow_pin.py -m gpu -q --timeout 1500 2>&1 | tail -15

#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests/test_gpu_parity.py tests/test_environment_light.py tests/test_golden.py -m gpu -x -q --timeout 1200 2>&1 | tail -8
tools/gpu_ab.sh "manylight cornell" hikari.jl_amd/csrc/libhikari_mi355x.so hikari.jl_amd/csrc/libhikari_prev.so 1

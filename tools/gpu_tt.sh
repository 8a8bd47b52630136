#!/bin/bash
# run selected gpu tests without stopping at the first failure:  tools/gpu_tt.sh "<pytest args>"
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 2400 python -m pytest $1 -m gpu -q --timeout 1200 -rf 2>&1 | tee gpurun_out/tt.log | tail -60

#!/bin/bash
# run selected gpu tests:  tools/gpu_t.sh "<pytest args>"
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest $1 -m gpu -x -q --timeout 1200 2>&1 | tail -25

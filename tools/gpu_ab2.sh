#!/bin/bash
# bitwise film compare of two builds, then bench configs under several:  tools/gpu_ab2.sh <libA> <libB> "<configs>" "<libs>" [repeats]
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python tools/ab_bitwise.py $1 $2 2>&1 | tail -3
tools/gpu_ab3.sh "$3" "$4" ${5:-1}

#!/bin/bash
# bitwise film compare of two builds, then the bench configs under both:  tools/gpu_ab2.sh <libA> <libB> "<configs>" [repeats]
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python tools/ab_bitwise.py $1 $2 2>&1 | tail -8
tools/gpu_ab.sh "$3" $1 $2 ${4:-1}

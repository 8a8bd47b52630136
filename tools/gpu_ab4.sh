#!/bin/bash
# utilisation probes of a debug build, bitwise film compare of two builds, then bench configs under several builds
#   tools/gpu_ab4.sh <dbglib|-> <libA> <libB> "<configs>" "<libs>" [repeats]
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
if [ "$1" != "-" ]; then tools/gpu_util.sh $1 cornell | grep HK_DEBUG_UTIL | head -6; fi
timeout 900 python tools/ab_bitwise.py $2 $3 2>&1 | tail -3
tools/gpu_ab3.sh "$4" "$5" ${6:-1}

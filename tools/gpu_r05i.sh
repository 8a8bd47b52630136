#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 600 python tools/ab_bitwise.py - scratch/lib_unroll2.so 2>&1 | tail -2
tools/gpu_ab5.sh "cornell manylight" "- scratch/lib_unroll2.so scratch/lib_unroll3.so" 2
tools/gpu_spp5.sh

#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python tools/ab_bitwise.py hikari.jl_amd/csrc/libhikari_before.so hikari.jl_amd/csrc/libhikari_mi355x.so 2>&1 | tail -12
tools/gpu_ab.sh "cloud" hikari.jl_amd/csrc/libhikari_mi355x.so hikari.jl_amd/csrc/libhikari_before.so 1

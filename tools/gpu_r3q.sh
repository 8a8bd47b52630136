#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
tools/gpu_ab.sh "cornell sky manylight" hikari.jl_amd/csrc/libhikari_mi355x.so hikari.jl_amd/csrc/libhikari_nopark.so 1
export HK_SOBOL_TABLE_ONLY=0
tools/gpu_ab.sh "cornell sky manylight" hikari.jl_amd/csrc/libhikari_mi355x.so hikari.jl_amd/csrc/libhikari_nopark.so 1

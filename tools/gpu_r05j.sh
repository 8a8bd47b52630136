#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
AB_ARGS="--spp 32 --steps 8" tools/gpu_ab5.sh cloud "- HK_DYNAMIC_SEGMENTS=0 HK_WAVES_PER_CU=48 HK_WAVES_PER_CU=48,HK_DYNAMIC_SEGMENTS=0 HK_WAVES_PER_CU=24 HK_WAVES_PER_CU=24,HK_DYNAMIC_SEGMENTS=0 HK_WAVES_PER_CU=192" 1
tools/gpu_ab5.sh "cornell manylight" "- scratch/lib_unroll4.so HK_BVH_LEAF=2 HK_BVH_LEAF=3 HK_BVH_LEAF=6" 1

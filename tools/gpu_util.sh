#!/bin/bash
# lane-slot bookkeeping of the state-machine kernels (a -DHK_DEBUG_UTIL build of the library given as $1):  tools/gpu_util.sh <lib> <config>
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
HK_OVERLAP=0 HK_LIB_PATH=$1 timeout 900 python bench.py --config ${2:-cornell} --no-cpu-baseline --warmup 0 --steps 2 2>&1 | grep -E "HK_DEBUG_UTIL|metric" | cut -c1-200

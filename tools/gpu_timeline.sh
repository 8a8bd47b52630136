#!/bin/bash
# kernel timeline of the LAST frame of a short bench run:   tools/gpu_timeline.sh <tag> <config> <spp> [n kernels]
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/trace -- python3 bench.py --config $2 --no-cpu-baseline --warmup 1 --steps 2 --spp $3 --progressive 0 --no-extra-configs --no-readback-pass > $O/trace.log 2>&1
python3 tools/rocpd_timeline.py $O/trace/*/*_results.db ${4:-40} > $O/timeline_$2_$3.txt 2>&1
find $O -name "*_results.db" -delete
cat $O/timeline_$2_$3.txt

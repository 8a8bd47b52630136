#!/bin/bash
# per-bounce kernel time of the last pass of a short bench run:   tools/gpu_depth.sh <tag> <config> <spp>
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/trace -- python3 bench.py --config $2 --no-cpu-baseline --warmup 1 --steps 1 --spp $3 --progressive 0 --no-extra-configs --no-readback-pass > $O/trace_$2.log 2>&1
python3 tools/rocpd_by_depth.py $O/trace/*/*_results.db > $O/by_depth_$2_$3.txt 2>&1
find $O -name "*_results.db" -delete
rm -rf $O/trace
cat $O/by_depth_$2_$3.txt

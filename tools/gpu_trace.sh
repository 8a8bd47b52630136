#!/bin/bash
# kernel trace of one config:  tools/gpu_trace.sh <tag> <config>
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; mkdir -p $O
export HK_OVERLAP=0
rocprofv3 --kernel-trace --stats -d $O/trace -- python3 bench.py --config $2 --no-cpu-baseline --steps 1 --warmup 1 > $O/trace.log 2>&1
python3 tools/rocpd_summary.py $O/trace/*/*_results.db > $O/trace_$2.txt 2>&1
find $O -name "*_results.db" -delete
head -12 $O/trace_$2.txt | cut -c1-165

#!/bin/bash
# kernel timeline of the one-sample calls rendered at once (batching off), Cornell 800^2: the last calls of bench.py's progressive loop
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r05n; mkdir -p $O
export HK_BATCH_PATHS_M=0
rocprofv3 --kernel-trace -d $O/trace -- python3 bench.py --config ${1:-cornell} --no-cpu-baseline --warmup 0 --steps 1 --spp 8 --progressive 64 --no-extra-configs --no-readback-pass > $O/trace.log 2>&1
tail -2 $O/trace.log | cut -c1-600
python3 tools/rocpd_timeline.py $O/trace/*/*_results.db ${2:-130} > $O/timeline_progressive.txt 2>&1
find $O -name "*_results.db" -delete
cat $O/timeline_progressive.txt | cut -c1-150

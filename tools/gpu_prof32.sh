#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/prof32; mkdir -p $O
export HK_OVERLAP=0
timeout 600 rocprofv3 --kernel-trace --stats -d $O/k -- python3 bench.py --no-cpu-baseline --spp ${1:-32} --steps 20 --progressive 0 --no-extra-configs > $O/log.txt 2>&1
python3 tools/rocpd_summary.py $O/k/*/*_results.db > $O/stats_spp${1:-32}.txt 2>&1
find $O -name "*_results.db" -delete
head -20 $O/stats_spp${1:-32}.txt | cut -c1-150
tail -1 $O/log.txt | cut -c1-300

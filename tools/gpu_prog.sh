#!/bin/bash
# the one-sample-per-call path: bench lines' `progressive` object for some configs, and a kernel trace of 32 one-sample calls
#   tools/gpu_prog.sh <tag> "<configs>" [lib]
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; mkdir -p $O
if [ -n "$3" ]; then export HK_LIB_PATH=$3; fi
for c in $2; do
  timeout 900 python bench.py --config $c --no-cpu-baseline --warmup 1 --steps 1 --progressive 64 2>$O/prog_$c.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$c', d['seconds_per_frame'], d.get('progressive'))"
done
if [ "$4" == "trace" ]; then
c=${2%% *}
rocprofv3 --kernel-trace --stats -d $O/trace_prog -- python3 bench.py --config $c --no-cpu-baseline --warmup 0 --steps 1 --spp 1 --progressive 32 --no-readback-pass > $O/trace_prog.log 2>&1
python3 tools/rocpd_summary.py $O/trace_prog/*/*_results.db > $O/trace_prog_$c.txt 2>&1
python3 tools/rocpd_timeline.py $O/trace_prog/*/*_results.db 50 > $O/timeline_prog_$c.txt 2>&1
find $O -name "*_results.db" -delete
cat $O/timeline_prog_$c.txt
fi

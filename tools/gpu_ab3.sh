#!/bin/bash
# bench configs under several library builds:  tools/gpu_ab3.sh "<configs>" "<libs>" [repeats]
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for c in $1; do for rep in $(seq 1 ${3:-1}); do for lib in $2; do
  HK_LIB_PATH=$lib timeout 900 python bench.py --config $c --no-cpu-baseline --warmup 1 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); ks=d['roofline']['kernel_seconds']; n=d['steps']
print('$c $lib', d['value'], d['seconds_per_frame'], {k: round(x/n,5) for k,x in ks.items()})"
done; done; done

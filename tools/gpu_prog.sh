#!/bin/bash
# the one-sample-per-call figures under several specs, interleaved on ONE box:  tools/gpu_prog.sh "<configs>" "<specs>" [repeats]
#   spec = - (defaults) | lib.so | VAR=v[,VAR=v...] | lib.so,VAR=v
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for c in $1; do for rep in $(seq 1 ${3:-1}); do for spec in $2; do
  envs=""; for part in ${spec//,/ }; do case $part in *=*) envs="$envs $part";; -) ;; *) envs="$envs HK_LIB_PATH=$part";; esac; done
  env $envs timeout 900 python bench.py --config $c --no-cpu-baseline --progressive 64 --warmup 0 --steps 1 --spp 8 --no-extra-configs --detail-file /tmp/pg5.json > /dev/null 2>/tmp/pg5.err || tail -3 /tmp/pg5.err
  python3 -c "
import json
p=json.load(open('/tmp/pg5.json'))['progressive']
print('$c $spec', {k: p[k] for k in ('ms_per_call','ms_per_call_with_readback','ms_per_call_pipelined_readback','ms_per_call_batched_no_readback')})"
done; done; done

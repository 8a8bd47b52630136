#!/bin/bash
# bench configs under several specs, interleaved on ONE box:  tools/gpu_ab5.sh "<configs>" "<specs>" [repeats]
#   AB_ARGS: extra bench.py arguments (e.g. "--spp 32 --steps 8")
#   spec = - (defaults) | lib.so | VAR=v[,VAR=v...] | lib.so,VAR=v
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for c in $1; do for rep in $(seq 1 ${3:-1}); do for spec in $2; do
  envs=""; for part in ${spec//,/ }; do case $part in *=*) envs="$envs $part";; -) ;; *) envs="$envs HK_LIB_PATH=$part";; esac; done
  env $envs timeout 900 python bench.py --config $c --no-cpu-baseline --progressive 0 --warmup 1 $AB_ARGS --detail-file /tmp/ab5.json > /dev/null 2>/tmp/ab5.err || tail -3 /tmp/ab5.err
  python3 -c "
import json
d=json.load(open('/tmp/ab5.json')); ks=d['roofline']['kernel_seconds']; n=d['steps']
print('$c $spec', d['seconds_per_frame'], d['value'], {k: round(x/n,5) for k,x in ks.items()})"
done; done; done

#!/bin/bash
# bitwise film compare of two environment / library specs (tools/ab_bitwise.py), then bench configs under several specs, interleaved on ONE box
#   tools/gpu_env_ab.sh <specA> <specB> "<configs>" "<specs>" [repeats]        spec = lib.so | VAR=v[,VAR=v...] | - (defaults)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
if [ "$1" != "-" ] || [ "$2" != "-" ]; then timeout 900 python tools/ab_bitwise.py "$1" "$2" 2>&1 | tail -12; fi
for c in $3; do for rep in $(seq 1 ${5:-1}); do for spec in $4; do
  envs=""; for part in ${spec//,/ }; do case $part in *=*) envs="$envs $part";; -) ;; *) envs="$envs HK_LIB_PATH=$part";; esac; done
  env $envs timeout 900 python bench.py --config $c --no-cpu-baseline --warmup 1 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); ks=d['roofline']['kernel_seconds']; n=d['steps']
print('$c $spec', d['value'], d['seconds_per_frame'], {k: round(x/n,5) for k,x in ks.items()})"
done; done; done

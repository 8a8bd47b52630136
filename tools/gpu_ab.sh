#!/bin/bash
# bench configs under several specs, interleaved on ONE box:  tools/gpu_ab.sh "<configs>" "<specs>" [repeats]
#   AB_ARGS: extra bench.py arguments (e.g. "--spp 32 --steps 8")
#   spec = - (defaults) | lib.so | VAR=v[,VAR=v...] | lib.so,VAR=v
# The specs run in the given order in odd repeats and in REVERSE order in even ones (A B | B A | A B ...): round 6 found that a process's
# frame time depends on its predecessor's — path-state placement alternates between two modes, 103.5 / 105.7 ms for the very same Cornell
# build in strict A B A B order — so a fixed order charges one spec the slow mode every time.  Use an even number of repeats.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for c in $1; do for rep in $(seq 1 ${3:-1}); do
  specs="$2"; if [ $((rep % 2)) -eq 0 ]; then specs=$(echo $2 | tr ' ' '\n' | tac | tr '\n' ' '); fi
  for spec in $specs; do
  envs=""; for part in ${spec//,/ }; do case $part in *=*) envs="$envs $part";; -) ;; *) envs="$envs HK_LIB_PATH=$part";; esac; done
  env $envs timeout 900 python bench.py --config $c --no-cpu-baseline --progressive 0 --warmup 1 $AB_ARGS --detail-file /tmp/ab5.json > /dev/null 2>/tmp/ab5.err || tail -3 /tmp/ab5.err
  python3 -c "
import json
d=json.load(open('/tmp/ab5.json')); ks=d['roofline']['kernel_seconds']; n=d['steps']
print('$c $spec', d['seconds_per_frame'], d['value'], {k: round(x/n,5) for k,x in ks.items()})"
done; done; done

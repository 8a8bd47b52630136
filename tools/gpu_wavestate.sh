#!/bin/bash
# Where a resident wave's time goes, per kernel:  tools/gpu_wavestate.sh <tag> <config> [...]
#   SQ_WAIT_ANY (parked at s_waitcnt / barrier: memory latency), SQ_WAIT_INST_ANY (issue stall), SQ_ACTIVE_INST_ANY (issuing) — disjoint,
#   their sum ~ SQ_WAVE_CYCLES (MI355X_MICROARCH.md, SQ counter table).  One counter group per pass.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
R=$1; shift
O=gpurun_out/$R; mkdir -p $O
export HK_OVERLAP=0
for CFG in "$@"; do
  for grp in "wait:SQ_WAIT_ANY SQ_WAVE_CYCLES" "act:SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY"; do
    name=ws_${grp%%:*}_$CFG
    timeout 900 rocprofv3 --kernel-trace --pmc ${grp#*:} -d $O/$name -- python3 bench.py --config $CFG --no-cpu-baseline --warmup 1 --steps 1 --progressive 0 --no-extra-configs --no-readback-pass > $O/$name.log 2>&1
    python3 tools/rocpd_summary.py $O/$name/*/*_results.db > $O/${R}_${name}.txt 2>&1
    find $O/$name -name "*_results.db" -delete
  done
  python3 tools/wavestate.py $O/${R}_ws_wait_$CFG.txt $O/${R}_ws_act_$CFG.txt > $O/${R}_wavestate_$CFG.txt 2>&1
  cat $O/${R}_wavestate_$CFG.txt
done

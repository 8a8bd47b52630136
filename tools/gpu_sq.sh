#!/bin/bash
# SQ utilisation counters of one config:  tools/gpu_sq.sh <tag> <config>
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; mkdir -p $O; CFG=$2; R=$1
export HK_OVERLAP=0
prof() {
  local name=$1; shift
  timeout 900 rocprofv3 --kernel-trace --pmc $* -d $O/$name -- python3 bench.py --config $CFG --no-cpu-baseline --warmup 1 --steps 1 > $O/$name.log 2>&1
  python3 tools/rocpd_summary.py $O/$name/*/*_results.db > $O/${R}_${name}_$CFG.txt 2>&1
}
prof sq_valu SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU
prof sq_busy SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU
prof sq_mem SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_ANY
prof grbm GRBM_GUI_ACTIVE SQ_WAVES
find $O -name "*_results.db" -delete
python3 tools/pmc_utilisation.py $O $R $CFG > $O/utilisation_$CFG.json
cat $O/utilisation_$CFG.json

#!/bin/bash
# k_small_pass against the launches it replaces: kernel durations and SQ counters (separate passes)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r05v; mkdir -p $O
prof() { local name=$1; shift; local pmc=""; if [ $# -gt 0 ]; then pmc="--pmc $*"; fi
  timeout 600 rocprofv3 --kernel-trace $pmc -d $O/$name -- python3 scratch/fused_prof.py > $O/$name.log 2>&1
  python3 tools/rocpd_summary.py $O/$name/*/*_results.db > $O/r05_small_pass_$name.txt 2>&1; }
prof kernels
prof sq_valu SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU
prof sq_busy SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU
prof sq_mem SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_ANY
find $O -name "*_results.db" -delete; find $O -type d -empty -delete
head -30 $O/r05_small_pass_kernels.txt | cut -c1-200
for f in sq_valu sq_busy sq_mem; do grep -A40 "counter" $O/r05_small_pass_$f.txt | grep -E "k_small_pass|k_shade|k_trace_lean|k_shadow|k_camera" | cut -c1-200; done

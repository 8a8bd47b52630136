#!/bin/bash
# Round evidence on the MI355X box (run through gpurun): bench lines of every config, rocprofv3 kernel stats, PMC traffic
# (FETCH_SIZE / WRITE_SIZE / L2 hit-miss in SEPARATE passes, as /opt/skills/guides/MI355X_MICROARCH.md prescribes) and the SQ / TA
# utilisation counters behind DESIGN.md's "what bounds the kernels".  Everything lands in gpurun_out/<round>/; the summaries
# (text / json, no databases) are then copied into profiles/ by hand.
#   tools/profile_round.sh r03 [quick|full|bench] ["cornell sky cloud manylight"]     (third argument: only these configs; bench: the bench records only)
R=${1:-r03}; MODE=$2; CFGS=${3:-cornell sky cloud manylight}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/$R; mkdir -p $O
if [ "$MODE" != "quick" ]; then
  # the line the driver records (Cornell + one warm frame of every other config + the one-sample-per-call path), then one line per config
  timeout 1200 python bench.py --detail-file $O/bench_${R}_default_detail.json > $O/bench_${R}_default.json 2> $O/bench_default.err
  for c in $CFGS; do
    timeout 900 python bench.py --config $c --detail-file $O/bench_${R}_${c}_detail.json > $O/bench_${R}_$c.json 2> $O/bench_$c.err
  done
fi
if [ "$MODE" = "bench" ]; then exit 0; fi
# the profiled runs keep every kernel on one stream (HK_OVERLAP=0): a kernel trace of overlapping kernels charges each of them the
# time it shared, and the per-kernel averages would no longer be comparable with the bench line's serial HIP-event replay
export HK_OVERLAP=0
prof() {  # name, config, extra bench args, counters...
  local name=$1 cfg=$2 extra=$3; shift 3
  local pmc=""; if [ $# -gt 0 ]; then pmc="--pmc $*"; fi
  local stats="--stats"; if [ $# -gt 0 ]; then stats=""; fi
  timeout 900 rocprofv3 --kernel-trace $stats $pmc -d $O/$name -- python3 bench.py --config $cfg --no-cpu-baseline --progressive 0 $extra > $O/$name.log 2>&1
  python3 tools/rocpd_summary.py $O/$name/*/*_results.db > $O/${R}_$name.txt 2>&1
}
for cfg in $CFGS; do
  case $cfg in
    cornell) prof kernel_stats_cornell800 cornell "" ;;
    cloud) prof kernel_stats_cloud1024 cloud "--warmup 1" ;;
    manylight) prof kernel_stats_manylight1024 manylight "--warmup 1" ;;
    sky) prof kernel_stats_sky800 sky "" ;;
  esac
done
for cfg in $CFGS; do
  extra="--warmup 1"
  prof pmc_fetch_$cfg $cfg "$extra" FETCH_SIZE
  prof pmc_write_$cfg $cfg "$extra" WRITE_SIZE
  prof pmc_l2_$cfg $cfg "$extra" TCC_HIT_sum TCC_MISS_sum
  python3 tools/pmc_traffic.py $O/pmc_fetch_$cfg/*/*_results.db $O/pmc_write_$cfg/*/*_results.db $O/pmc_l2_$cfg/*/*_results.db \
      "bench.py --config $cfg $extra; round ${R}." > $O/pmc_traffic_$cfg.json
  prof sq_valu_$cfg $cfg "$extra" SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU
  prof sq_busy_$cfg $cfg "$extra" SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU
  prof sq_mem_$cfg $cfg "$extra" SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_ANY
  prof grbm_$cfg $cfg "$extra" GRBM_GUI_ACTIVE SQ_WAVES
  python3 tools/pmc_utilisation.py $O $R $cfg > $O/utilisation_$cfg.json
done
find $O -name "*_results.db" -delete
find $O -type d -empty -delete
ls $O

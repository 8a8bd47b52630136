#!/bin/bash
# the final bench lines of round 5 (after the counters of tools/profile_round.sh r05 were copied into profiles/)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r05l; mkdir -p $O
timeout 1200 python bench.py --detail-file $O/bench_r05_default_detail.json > $O/bench_r05_default.json 2> $O/bench_default.err; echo rc=$?
for c in cornell sky cloud manylight; do
  timeout 900 python bench.py --config $c --detail-file $O/bench_r05_${c}_detail.json > $O/bench_r05_$c.json 2> $O/bench_$c.err; echo $c rc=$?
done
wc -c $O/bench_r05_default.json
timeout 900 python -m pytest tests/test_c_abi.py tests/test_multi_gpu.py -m gpu -q --timeout 800 2>&1 | tail -3

#!/bin/bash
# The four BASELINE configs through bench.py, one JSON line each into gpurun_out/<round>/bench_<round>_<config>.json (copy to profiles/).
R=${1:-rXX}
O=gpurun_out/$R
mkdir -p $O
for c in cornell sky cloud manylight; do
  python3 bench.py --config $c > $O/bench_${R}_$c.json 2> $O/bench_$c.err || echo "bench $c failed"
  tail -c 400 $O/bench_${R}_$c.json
done

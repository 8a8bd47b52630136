#!/bin/bash
# lane-slot bookkeeping of the surface traversal kernels (probes 0-2, 6, 8: k_trace_lean; 3-5, 7: k_shadow) in the Cornell box and the 10^6-triangle scene
for c in cornell manylight; do echo "== $c"; tools/gpu_util.sh scratch/lib_dbg.so $c; done

#!/bin/bash
# round 5: which change broke test_full_size_many_light (pool kernel? pair layout? second stream?), the external-accumulator test in its own process
mkdir -p gpurun_out
T="tests/test_parity_holes.py::test_full_size_many_light"
for spec in "HK_SELECT_POOL=1" "HK_SELECT_POOL=0" "HK_SELECT_POOL=1 HK_OVERLAP=0" "HK_PRESELECT=0"; do
  echo "== $spec"
  env $spec timeout 600 python -m pytest $T -m gpu -q --timeout 500 2>&1 | grep -E "AssertionError:|passed|failed" | head -5
done
timeout 600 python -m pytest "tests/test_gpu_parity.py::test_small_calls_into_external_accumulators_are_stream_ordered" "tests/test_gpu_parity.py::test_light_bvh_parity" -m gpu -q --timeout 500 2>&1 | tail -15

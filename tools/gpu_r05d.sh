#!/bin/bash
# round 5: the r05b sequence again (order-dependent failure of test_full_size_many_light?), under both select kernels
mkdir -p gpurun_out
SEQ='tests/test_gpu_parity.py::test_light_bvh_parity tests/test_parity_holes.py::test_light_preselection_is_result_neutral tests/test_parity_holes.py::test_scheduling_is_result_neutral tests/test_gpu_parity.py::test_frame_parity tests/test_parity_holes.py::test_full_size_many_light'
for spec in "HK_SELECT_POOL=1" "HK_SELECT_POOL=0"; do
  echo "== $spec"
  env $spec timeout 900 python -m pytest $SEQ -m gpu -q --timeout 800 -p no:randomly 2>&1 | grep -E "AssertionError|passed|failed|FAILED" | head -8
done
echo "== preselect + full only"
timeout 900 python -m pytest tests/test_parity_holes.py::test_light_preselection_is_result_neutral tests/test_parity_holes.py::test_full_size_many_light -m gpu -q --timeout 800 2>&1 | grep -E "AssertionError|passed|failed|FAILED" | head -8
echo "== scheduling + full only"
timeout 900 python -m pytest tests/test_parity_holes.py::test_scheduling_is_result_neutral tests/test_parity_holes.py::test_full_size_many_light -m gpu -q --timeout 800 2>&1 | grep -E "AssertionError|passed|failed|FAILED" | head -8

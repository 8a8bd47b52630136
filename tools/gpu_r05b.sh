#!/bin/bash
# round 5: light-BVH pair layout + pooled k_light_select: parity, neutrality, the failed tests of r05a again, the many-light bench line
mkdir -p gpurun_out
timeout 1500 python -m pytest "tests/test_gpu_parity.py::test_light_bvh_parity" "tests/test_gpu_parity.py::test_small_calls_into_external_accumulators_are_stream_ordered" "tests/test_parity_holes.py::test_light_preselection_is_result_neutral" "tests/test_parity_holes.py::test_scheduling_is_result_neutral" "tests/test_gpu_parity.py::test_frame_parity" "tests/test_parity_holes.py::test_full_size_many_light" tests/test_control_flow_pin.py -m gpu -q --timeout 900 2>&1 | tail -60 > gpurun_out/r05b_tests.log
cat gpurun_out/r05b_tests.log
for pool in 1 0; do
HK_SELECT_POOL=$pool timeout 600 python bench.py --config manylight --no-cpu-baseline --progressive 0 --detail-file gpurun_out/r05b_manylight_pool$pool.json > gpurun_out/r05b_manylight_pool$pool.out 2> gpurun_out/r05b_manylight_pool$pool.err; echo bench rc=$?
python - <<P
import json
d=json.load(open("gpurun_out/r05b_manylight_pool$pool.json"))
print("pool=$pool", d["seconds_per_frame"], d["value"], d["roofline"]["kernel_seconds"])
P
done

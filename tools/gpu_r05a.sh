#!/bin/bash
# round 5, first GPU call: the tests touched by the knob / read-back / ordering changes, then the default bench line
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_control_flow_pin.py "tests/test_parity_holes.py::test_scheduling_is_result_neutral" "tests/test_parity_holes.py::test_lean_traversal_parity" "tests/test_parity_holes.py::test_zsobol_sample_bit_table" "tests/test_parity_holes.py::test_light_preselection_is_result_neutral" "tests/test_parity_holes.py::test_node_cache_partial_tree" "tests/test_converged_parity.py::test_bomex_crop_converged_parity" tests/test_abi_errors.py -m gpu -q --timeout 900 2>&1 | tail -25 > gpurun_out/r05a_tests.log
cat gpurun_out/r05a_tests.log
timeout 600 python bench.py > gpurun_out/r05a_bench.out 2> gpurun_out/r05a_bench.err; echo bench rc=$?
cp bench_detail.json gpurun_out/r05a_bench_detail.json
tail -c 3600 gpurun_out/r05a_bench.out

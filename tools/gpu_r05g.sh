#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_control_flow_pin.py tests/test_parity_holes.py::test_light_preselection_is_result_neutral "tests/test_gpu_parity.py::test_readback_every_call_and_pipelined" "tests/test_gpu_parity.py::test_lane_pipeline_rebuilds_the_sample_bit_table_behind_the_lanes" tests/test_gpu_parity.py::test_context_options -m gpu -q --timeout 800 2>&1 | tail -8
tools/gpu_ab5.sh manylight "- scratch/lib_sel256n256w5.so scratch/lib_sel512n1024.so" 2
tools/gpu_ab5.sh cloud "- scratch/lib_walk3.so" 2

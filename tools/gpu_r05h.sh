#!/bin/bash
# postponed leaves in the lean traversal kernels: parity, bit-identity against the build without them, the three surface configs A/B
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_parity_holes.py::test_lean_traversal_parity tests/test_parity_holes.py::test_node_cache_partial_tree "tests/test_gpu_parity.py::test_frame_parity" tests/test_gpu_parity.py::test_closest_hit_parity_1M_rays -m gpu -q --timeout 800 2>&1 | tail -4
timeout 600 python tools/ab_bitwise.py - scratch/lib_post00.so 2>&1 | tail -12
tools/gpu_ab5.sh "cornell manylight sky" "- scratch/lib_post00.so scratch/lib_post10.so scratch/lib_post01.so" 2

#!/bin/bash
# last check of a round: smoke, the default bench line, a slice of the GPU suite
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python bench.py 2>/dev/null | tail -1 | cut -c1-600
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_abi_errors.py -m gpu -q 2>&1 | tail -2

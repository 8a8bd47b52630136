#!/bin/bash
# exploratory: the STRICT random-scene comparison at ~280^2 (every wave segment refilled several times) on seeds the suite does not contain:
#   tools/gpu_fuzz_large.sh <first> <count>
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 2400 python - <<PY
import sys
sys.path[:0] = [".", "oracle", "tests"]
import numpy as np, hikari_jl_amd as hk, oracle
import test_fuzz_parity as T
first, count = $1, $2
bad = n = 0
for klass, size in (("closed", (311, 257)), ("absorbing", (256, 300)), ("wild", (283, 277))):
    for seed in range(first, first + count):
        n += 1
        try:
            T.test_fuzz_strict(hk, oracle, klass, seed, size)
        except AssertionError as e:
            bad += 1
            print("FAIL", klass, seed, str(e)[:400], flush=True)
        except Exception as e:
            bad += 1
            print("ERROR", klass, seed, repr(e)[:400], flush=True)
print("done: %d failures of %d large scenes" % (bad, n))
PY

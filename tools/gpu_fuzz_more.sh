#!/bin/bash
# exploratory: the random-scene comparison of tests/test_fuzz_parity.py on seeds the suite does not contain:  tools/gpu_fuzz_more.sh <first> <count>
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 2400 python - <<PY
import sys, traceback
sys.path[:0] = [".", "oracle", "tests"]
import numpy as np, hikari_jl_amd as hk, oracle
import test_fuzz_parity as T
first, count = $1, $2
bad = 0
for klass, fn in (("closed", T.test_fuzz_strict), ("absorbing", T.test_fuzz_strict), ("wild", T.test_fuzz_strict), ("walk", T.test_fuzz_converged), ("scatter", T.test_fuzz_converged), ("wild_scatter", T.test_fuzz_converged)):
    for seed in range(first, first + count):
        try:
            if fn is T.test_fuzz_strict:
                fn.__wrapped__(hk, oracle, klass, seed, None) if hasattr(fn, "__wrapped__") else fn(hk, oracle, klass, seed, None)
            else:
                fn(hk, oracle, klass, seed)
        except AssertionError as e:
            bad += 1
            print("FAIL", klass, seed, str(e)[:400])
        except Exception as e:
            bad += 1
            print("ERROR", klass, seed, repr(e)[:400])
print("done: %d failures of %d scenes" % (bad, 6 * count))
PY

#!/usr/bin/env python3
"""Null distribution of the random-scene converged bar (tests/test_fuzz_parity.py::test_fuzz_converged): the ORACLE against ITSELF.

    python tools/fuzz_null.py <class> <first seed> <count> [--old]        (CPU only; ~3 s per scene on 8 cores)

For each random scene: A = the oracle's 512 spp in 8 batches (sample indices 1 .. 512), B = the oracle's 2 048 OTHER spp in 8 batches
(513 .. 2 560) — what the device renders in the test.  Both estimate the same image, so every statistic printed here is what a CORRECT
device would score; the test's bounds are set from these (the one-sample bar of rounds 3-5 — z against the
variance of A's eight batches alone, |z| > 6 on <= 15 % of the lit channels — scored 16.2 % oracle-against-oracle on `wild_scatter
10000`: a bar with an unknown false-positive rate).  `--old` prints that statistic beside the two-sample one."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]


def two_sample_stats(FA, FB):
    """-> (median z^2, fraction |z| > 6) over the lit pixel channels; z = (mean B - mean A) / sqrt(se_A^2 + se_B^2 + floor^2), both standard
    errors from the batch scatter of their own side"""
    A, B = FA.mean(axis=0), FB.mean(axis=0)
    seA = FA.std(axis=0, ddof=1) / np.sqrt(FA.shape[0])
    seB = FB.std(axis=0, ddof=1) / np.sqrt(FB.shape[0])
    z = (B - A) / np.sqrt(seA ** 2 + seB ** 2 + (1e-4 * (A + 1e-3)) ** 2)
    lit = (A.sum(axis=2) > 0) | (B.sum(axis=2) > 0)
    if lit.sum() < 16:
        return 0.0, 0.0, int(lit.sum())
    return float(np.median((z ** 2)[lit])), float(np.mean(np.abs(z[lit]) > 6.0)), int(lit.sum())


def main():
    import hikari_jl_amd as hk
    import oracle
    import test_fuzz_parity as T
    klass, first, count = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    old = "--old" in sys.argv
    oracle.build()
    rows = []
    for seed in range(first, first + count):
        s, film, cam, kw, desc = T.random_scene(hk, seed, klass, (16, 16))
        kw = {k: v for k, v in kw.items() if k != "samples"}
        kw["samples"] = 512 + 2048
        p = hk.integrator_params(**kw)
        osc = oracle.OracleScene(s)
        FA = np.stack([oracle.finalize(osc.render(p, cam, 16, 16, 64, first=1 + 64 * b)[0], 16, 16) for b in range(8)])
        FB = np.stack([oracle.finalize(osc.render(p, cam, 16, 16, 256, first=513 + 256 * b)[0], 16, 16) for b in range(8)])
        osc.close()
        med, tail, n_lit = two_sample_stats(FA, FB)
        A, B = FA.mean(axis=0), FB.mean(axis=0)
        worst_mean = max(abs(B[..., c].mean() - A[..., c].mean()) / max(A[..., c].mean(), 1e-9) for c in range(3))
        line = "%s %d: lit %d  median z^2 %.3f  |z|>6 %.4f  channel means off by %.4f" % (klass, seed, n_lit, med, tail, worst_mean)
        if old:
            seA = FA.std(axis=0, ddof=1) / np.sqrt(8)
            z = (B - A) / np.sqrt(seA ** 2 * 1.25 + (1e-4 * (A + 1e-3)) ** 2)
            lit = A.sum(axis=2) > 0
            line += "   | one-sample bar of rounds 3-5: median z^2 %.3f  |z|>6 %.4f" % (float(np.median((z ** 2)[lit])), float(np.mean(np.abs(z[lit]) > 6.0)))
        print(line, flush=True)
        rows.append((med, tail, worst_mean))
    r = np.array(rows)
    print("%s seeds %d..%d: median z^2 max %.3f (mean %.3f)   |z|>6 max %.4f (mean %.4f)   means max %.4f" % (
        klass, first, first + count - 1, r[:, 0].max(), r[:, 0].mean(), r[:, 1].max(), r[:, 1].mean(), r[:, 2].max()))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Device-only robustness sweep: random scenes of tests/fuzz_scenes.py on a 16 x 16 film at MANY samples per pixel (default 4 096 — the
rare events a 64-spp parity run never meets: quirk Q35 was one NaN pixel in 2 560 samples) — every pixel finite and non-negative.
    tools/fuzz_finite.py <first seed> <count> [spp]        (run on the GPU box; prints the scenes that fail and a summary line)"""
import sys
import time
sys.path[:0] = [".", "tests"]
import numpy as np
import hikari_jl_amd as hk
from fuzz_scenes import random_scene

first, count = int(sys.argv[1]), int(sys.argv[2])
spp = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
bad, n, t0 = 0, 0, time.time()
for klass in ("closed", "absorbing", "walk", "scatter", "wild", "wild_scatter"):
    for seed in range(first, first + count):
        s, film, cam, kw, desc = random_scene(hk, seed, klass, (16, 16))
        kw = dict({k: v for k, v in kw.items() if k != "samples"}, samples=spp)
        film = hk.Film((16, 16))
        try:
            vp = hk.VolPath(**kw)
            vp(s, film, cam)
            fb = film.framebuffer
            ok = np.isfinite(fb).all() and (fb >= 0).all()
            vp.close()
        except Exception as e:      # a scene the library rejects is reported, not fatal
            ok = False
            desc = [repr(e)[:200]]
        n += 1
        if not ok:
            bad += 1
            print("FAIL", klass, seed, " ".join(desc)[:300], flush=True)
print("done: %d failures of %d scenes at %d spp in %.0f s" % (bad, n, spp, time.time() - t0))

#!/usr/bin/env python3
"""One random scene of tests/fuzz_scenes.py on both sides, with the counters the fuzz tests compare:
    tools/fuzz_seed.py <class> <seed>          (run on the GPU box)"""
import sys
sys.path[:0] = [".", "oracle", "tests"]
import numpy as np
import hikari_jl_amd as hk
import oracle
from fuzz_scenes import random_scene

klass, seed = sys.argv[1], int(sys.argv[2])
s, film, cam, kw, desc = random_scene(hk, seed, klass, None)
w, h = film.width, film.height
acc, ost = oracle.OracleScene(s).render(hk.integrator_params(**kw), cam, w, h, kw["samples"])
ref = oracle.finalize(acc, w, h)
vp = hk.VolPath(**kw)
vp(s, film, cam)
g = film.framebuffer.copy()
st = vp.stats()
vp.close()
rel = np.sqrt(((g - ref) ** 2).sum(axis=2)) / (np.sqrt((ref ** 2).sum(axis=2)) + 1e-6)
print(desc)
print("pixels beyond 1e-2: %d of %d; relMSE %.3g" % (int((rel > 1e-2).sum()), w * h, float(np.mean((g - ref) ** 2 / (ref ** 2 + 1e-3)))))
print("closest casts: device %d oracle %d; shadow casts: device %d oracle %d" % (st.rays_closest, ost.rays_closest, st.rays_shadow, ost.rays_shadow))

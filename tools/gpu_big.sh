#!/bin/bash
# large films: the pass sizing, the 32-bit slot arithmetic and the tables at 4K / 8K
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 1200 python - <<PY
import sys, time
sys.path.insert(0, ".")
import numpy as np, hikari_jl_amd as hk
from hikari_jl_amd import scenes
ref = None
for (w, h, spp) in ((256, 256, 64), (4096, 4096, 16), (7680, 4320, 4), (8192, 8192, 2), (16384, 1024, 4), (1000, 12000, 5)):
    s, film, cam = scenes.cornell_box(w, h, light="area")
    cam = hk.PerspectiveCamera((0, 1, -3.5), (0, 1, 0), film, fov=40.0, screen_window=((-w / h, -1), (w / h, 1)))
    vp = hk.VolPath(max_depth=8, samples=spp)
    t = time.time()
    vp(s, film, cam)
    dt = time.time() - t
    st = vp.stats()
    fb = film.framebuffer
    cy, cx = h // 2, w // 2
    centre = fb[cy - h // 8: cy + h // 8, cx - min(w, h) // 8: cx + min(w, h) // 8].mean()
    print("%5d x %5d  %3d spp  %.2f s (incl. scene upload + readback)  %.0f Mrays  finite %s  mean %.4f  centre %.4f" % (w, h, spp, dt, (st.rays_closest + st.rays_shadow) / 1e6, np.isfinite(fb).all(), fb.mean(), centre))
    vp.close()
PY

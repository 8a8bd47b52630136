#!/usr/bin/env python3
"""Film accumulators of a few small scenes under two builds of the library (HK_LIB_PATH), compared bit for bit: for changes that
only re-schedule work (loop shapes, kernel splits).   python tools/ab_bitwise.py <specA> <specB>
A spec is a library path, or comma-separated VAR=value settings (the shipped library under those environment variables), or both
joined by commas: "HK_GREY_FLAT=0", "hikari.jl_amd/csrc/libx.so,HK_GREY=0"."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, numpy as np
sys.path.insert(0, %(root)r)
import hikari_jl_amd as hk
from hikari_jl_amd import scenes
out = {}
cases = {"cloud_nanovdb": (lambda: scenes.cloud_scene(40, 36, "nanovdb", res=(48, 48, 24)), 10),
         "cloud_grid": (lambda: scenes.cloud_scene(40, 36, "grid", res=(48, 48, 24)), 10),
         "bomex": (lambda: scenes.bomex_scene(48, 48, res=(64, 64, 32)), 16),
         "integration": (lambda: scenes.integration_test_scene(40, 36), 5),
         "cornell": (lambda: scenes.cornell_box(40, 36, light="area"), 6)}
for name, (mk, depth) in cases.items():
    s, film, cam = mk()
    vp = hk.VolPath(max_depth=depth, samples=64)
    vp._ensure(film); vp.clear()
    vp.render_samples(s, film, cam, 24, first=1, readback=False)
    out[name] = vp.read_accumulators(film).copy()
    st = vp.stats()
    out[name + "_counts"] = np.array([st.rays_closest, st.rays_shadow, st.medium_collisions], np.int64)
    vp.close()
np.savez(sys.argv[1], **out)
'''


def main(lib_a, lib_b):
    import numpy as np
    outs = []
    for i, lib in enumerate((lib_a, lib_b)):
        path = "/tmp/ab_bitwise_%d.npz" % i
        env = dict(os.environ)
        for part in lib.split(","):
            if "=" in part:
                k, v = part.split("=", 1)
                env[k] = v
            elif part and part != "-":
                env["HK_LIB_PATH"] = os.path.abspath(part)
        subprocess.check_call([sys.executable, "-c", CHILD % {"root": ROOT}, path], env=env)
        outs.append(np.load(path))
    ok = True
    for k in outs[0].files:
        a, b = outs[0][k], outs[1][k]
        same = np.array_equal(a.view(np.uint8), b.view(np.uint8))
        ok &= same
        print("%-22s %s" % (k, "identical" if same else "DIFFERENT (max abs %.3g)" % float(np.abs(a.astype(np.float64) - b.astype(np.float64)).max())))
    print("ALL IDENTICAL" if ok else "DIFFERENCES")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main(sys.argv[1], sys.argv[2]))

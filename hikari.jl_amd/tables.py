"""Data tables the reference embeds in its sources, as binary fixtures (tools/extract_reference_tables.py,
tools/gen_rgb2spec.c).  They are handed to the C-ABI through hk_tables (include/hikari_mi355x.h)."""
import ctypes as C
import os
import subprocess

import numpy as np

from . import _abi as A

DATA_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")
_cache = {}


def _ensure_rgb2spec(path):
    """The reference regenerates srgb_spectrum_table.dat when it is missing (spectral/rgb2spec.jl:418-429)."""
    if os.path.isfile(path):
        return
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = os.path.join(root, "tools", "gen_rgb2spec.c")
    exe = os.path.join(root, "tools", "gen_rgb2spec")
    subprocess.check_call(["gcc", "-O2", "-fopenmp", "-o", exe, src, "-lm"])
    subprocess.check_call([exe, path, "64"])


def load():
    """-> dict(sobol=u32[1024*52], cie=(x,y,z) f32[471], res, scale f32[res], coeffs f32[...], struct=hk_tables)"""
    if "t" in _cache:
        return _cache["t"]
    sobol = np.fromfile(os.path.join(DATA_DIR, "sobol_matrices.bin"), dtype=np.uint32)
    cie = np.fromfile(os.path.join(DATA_DIR, "cie_xyz.bin"), dtype=np.float32)
    assert sobol.size == 1024 * 52 and cie.size == 3 * 471
    p = os.path.join(DATA_DIR, "srgb_spectrum_table.dat")
    if os.environ.get("HK_RGB2SPEC_TABLE"):      # another table in the reference's own file layout (rgb2spec.jl:403-421): the one a Julia box dumped beside its fixtures
        p = os.environ["HK_RGB2SPEC_TABLE"]      # (tests/test_reference_fixtures.py: both sides must look colours up in the SAME table)
    _ensure_rgb2spec(p)
    raw = np.fromfile(p, dtype=np.uint8)
    res = int(raw[:4].view(np.int32)[0])
    scale = raw[4:4 + 4 * res].view(np.float32).copy()
    coeffs = raw[4 + 4 * res:].view(np.float32).copy()
    assert coeffs.size == 3 * res ** 3 * 3
    cx, cy, cz = (np.ascontiguousarray(cie[i * 471:(i + 1) * 471]) for i in range(3))
    st = A.hk_tables()
    st.sobol_matrices = sobol.ctypes.data_as(C.POINTER(C.c_uint32))
    st.sobol_count = sobol.size
    st.rgb2spec_res = res
    st.cie_x = cx.ctypes.data_as(A.PF)
    st.cie_y = cy.ctypes.data_as(A.PF)
    st.cie_z = cz.ctypes.data_as(A.PF)
    st.rgb2spec_scale = scale.ctypes.data_as(A.PF)
    st.rgb2spec_coeffs = coeffs.ctypes.data_as(A.PF)
    t = dict(sobol=sobol, cie=(cx, cy, cz), res=res, scale=scale, coeffs=coeffs, struct=st)
    _cache["t"] = t
    return t


# ---- host-side sigmoid-polynomial lookup used by the RGB light constructors (spectral/rgb2spec.jl:85-167,
# 371-385): PointLight(rgb::RGB, ...) etc. bake an RGBIlluminantSpectrum on the host. -------------------------
def rgb_to_spectrum(r, g, b):
    t = load()
    f = np.float32
    res = t["res"]
    scale = t["scale"]
    co = t["coeffs"]
    r, g, b = (f(min(max(float(v), 0.0), 1.0)) for v in (r, g, b))
    if r == g and g == b:
        if 0 < r < 1:
            c2 = f((r - f(0.5)) / np.sqrt(f(r * (f(1) - r))))
        elif r <= 0:
            c2 = f(-1e10)
        else:
            c2 = f(1e10)
        return (f(0), f(0), c2)
    maxc = (1 if r > b else 3) if r > g else (2 if g > b else 3)
    z = (r, g, b)[maxc - 1]
    xc = {1: g, 2: b, 3: r}[maxc]
    yc = {1: b, 2: r, 3: g}[maxc]
    x = f(f(xc * f(res - 1)) / z)
    y = f(f(yc * f(res - 1)) / z)
    zi = 1
    for i in range(1, res):
        if scale[i - 1] < z:
            zi = i
    zi = min(zi, res - 1)
    xi = min(int(x) + 1, res - 1)
    yi = min(int(y) + 1, res - 1)
    dx = f(x - f(xi - 1))
    dy = f(y - f(yi - 1))
    dz = f(f(z - scale[zi - 1]) / f(scale[zi] - scale[zi - 1]))

    def at(mc, z_, y_, x_, c_):
        return co[(mc - 1) + 3 * ((z_ - 1) + res * ((y_ - 1) + res * ((x_ - 1) + res * (c_ - 1))))]

    out = []
    one = f(1)
    for k in (1, 2, 3):
        a = f(f(one - dx) * at(maxc, zi, yi, xi, k)) + f(dx * at(maxc, zi, yi, xi + 1, k))
        b_ = f(f(one - dx) * at(maxc, zi, yi + 1, xi, k)) + f(dx * at(maxc, zi, yi + 1, xi + 1, k))
        c = f(f(one - dx) * at(maxc, zi + 1, yi, xi, k)) + f(dx * at(maxc, zi + 1, yi, xi + 1, k))
        d = f(f(one - dx) * at(maxc, zi + 1, yi + 1, xi, k)) + f(dx * at(maxc, zi + 1, yi + 1, xi + 1, k))
        lo = f(f(f(one - dy) * f(a)) + f(dy * f(b_)))
        hi = f(f(f(one - dy) * f(c)) + f(dy * f(d)))
        out.append(f(f(f(one - dz) * lo) + f(dz * hi)))
    return tuple(out)

"""tests/golden: regression fixtures made by tests/golden/make_golden.py (the build's own oracle: the reference cannot run
here, "parity unpinned") and the SHA-256 of the numeric tables extracted verbatim from the reference's sources."""
import ctypes as C
import hashlib
import importlib.util
import json
import os

import numpy as np
import pytest

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _mk():
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(HERE, "make_golden.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


CASES = ["single_triangle", "cornell_area", "cornell_point", "coated_diffuse", "glass", "sky", "slab_homogeneous"]


def test_reference_data_tables_are_pinned():
    want = json.load(open(os.path.join(HERE, "data_tables.json")))
    for name, digest in want.items():
        data = open(os.path.join(ROOT, "hikari.jl_amd", "data", name), "rb").read()
        assert hashlib.sha256(data).hexdigest() == digest, name


def test_reference_data_tables_match_the_reference_source():
    """where the reference tree is present (the build container), the Sobol matrices and CIE tables are re-read from its sources"""
    ref = "/root/reference/src"
    if not os.path.isdir(ref):
        pytest.skip("reference tree not present")
    import re
    import extract_reference_tables as ex
    sob = ex.grab_block(open(os.path.join(ref, "sampler/sobol_matrices.jl")).read(), "SobolMatrices32")
    arr = np.array([int(t, 16) for t in sob], dtype=np.uint32)
    have = np.fromfile(os.path.join(ROOT, "hikari.jl_amd", "data", "sobol_matrices.bin"), dtype=np.uint32)
    assert np.array_equal(arr, have)
    col = open(os.path.join(ref, "spectral/color.jl")).read()
    xyz = np.fromfile(os.path.join(ROOT, "hikari.jl_amd", "data", "cie_xyz.bin"), dtype=np.float32)
    got = []
    for nm in ("CIE_X", "CIE_Y", "CIE_Z"):
        t = ex.grab_block(col, nm)
        got.append(np.array([float(re.sub(r"f0$", "", x)) for x in t], dtype=np.float32))
    got = np.concatenate(got)
    assert got.size <= xyz.size and np.array_equal(got, xyz[-got.size:] if xyz.size != got.size else xyz)


@pytest.mark.parametrize("name", CASES)
def test_oracle_reproduces_golden_frames(name):
    mk = _mk()
    import oracle
    oracle.build()
    g = np.load(os.path.join(HERE, name + ".npz"))
    img, counts = mk.render(name)
    assert np.array_equal(counts, g["counts"]), (counts, g["counts"])
    assert np.allclose(img, g["framebuffer"], rtol=1e-5, atol=1e-7)


def test_oracle_reproduces_kat_vectors():
    import oracle
    oracle.build()
    g = np.load(os.path.join(HERE, "kat_vectors.npz"))
    s1, s2 = oracle.sobol(800, 800, 256, 0, g["px"], g["py"], g["idx"], g["dim"])
    assert np.array_equal(s1, g["sobol_1d"]) and np.array_equal(s2, g["sobol_2d"])
    for mode, key in enumerate(("uplift_bounded", "uplift_unbounded", "uplift_illuminant")):
        assert np.allclose(oracle.uplift(mode, g["rgb"], g["lam"]), g[key], rtol=1e-6, atol=1e-7)


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_gpu_matches_golden_frames(hk, name):
    """the HIP path through the C-ABI against the committed frames: the frame-parity bar of SURVEY 8(d) (relMSE <= 1e-3 and
    >= 99 % of pixels within 1e-2); the medium and coated-walk cases are statistical (their RNG streams are seeded from the bit
    patterns of rays / directions, so one differing ulp re-rolls a walk): no farther from frame A than the committed independent frame B"""
    mk = _mk()
    build, kw, (w, h) = mk.cases()[name]
    scene, film, cam = build(w, h)
    g = np.load(os.path.join(HERE, name + ".npz"))
    vp = hk.VolPath(**kw)
    vp(scene, film, cam)
    st = vp.stats()
    vp.close()
    got, ref = film.framebuffer, g["framebuffer"]
    assert np.isfinite(got).all()
    if name in ("slab_homogeneous", "coated_diffuse"):   # RNG streams seeded from float bit patterns (media, coated random walks)
        # the committed frame B (the next sample indices, same oracle) is the yardstick: the GPU frame must be no farther from frame A
        # than an independent frame of the oracle itself is, and its mean no further off than B's
        b = g["framebuffer_b"]

        def dist(x, y):
            return float(np.mean((x - y) ** 2 / (0.25 * (x + y) ** 2 + 1e-2)))

        assert dist(got, ref) <= 1.5 * dist(b, ref) + 1e-4, (name, dist(got, ref), dist(b, ref))
        assert abs(got.mean() - ref.mean()) <= max(0.02 * ref.mean(), 1.5 * abs(b.mean() - ref.mean())), (name, got.mean(), ref.mean(), b.mean())
        return
    rel_mse = float(np.mean((got - ref) ** 2 / (ref ** 2 + 1e-3)))
    num = np.sqrt(((got - ref) ** 2).sum(axis=2))
    den = np.sqrt((ref ** 2).sum(axis=2)) + 1e-6
    assert rel_mse <= 1e-3 and float(np.mean(num / den <= 1e-2)) >= 0.99, (name, rel_mse)
    assert abs(int(st.rays_closest) - int(g["counts"][0])) <= 0.002 * g["counts"][0] + 8


@pytest.mark.gpu
def test_gpu_matches_kat_vectors(hk, gpu_ctx):
    g = np.load(os.path.join(HERE, "kat_vectors.npz"))
    n = len(g["px"])
    L = hk._lib.lib()
    PI = C.POINTER(C.c_int32)
    a = [np.ascontiguousarray(g[k], np.int32) for k in ("px", "py", "idx", "dim")]
    g1, g2 = np.empty(n, np.float32), np.empty((n, 2), np.float32)
    hk._lib.check(L.hk_test_sobol(gpu_ctx.h, 800, 800, 256, 0, n, a[0].ctypes.data_as(PI), a[1].ctypes.data_as(PI), a[2].ctypes.data_as(PI),
                                  a[3].ctypes.data_as(PI), g1.ctypes.data_as(hk._abi.PF), g2.ctypes.data_as(hk._abi.PF)), "hk_test_sobol")
    assert np.array_equal(g1, g["sobol_1d"]) and np.array_equal(g2, g["sobol_2d"])
    rgb, lam = np.ascontiguousarray(g["rgb"]), np.ascontiguousarray(g["lam"])
    for mode, key in enumerate(("uplift_bounded", "uplift_unbounded", "uplift_illuminant")):
        out = np.empty_like(lam)
        hk._lib.check(L.hk_test_uplift(gpu_ctx.h, mode, n, rgb.ctypes.data_as(hk._abi.PF), lam.ctypes.data_as(hk._abi.PF), out.ctypes.data_as(hk._abi.PF)), "hk_test_uplift")
        assert np.allclose(out, g[key], rtol=3e-7 * 4, atol=1e-7), key

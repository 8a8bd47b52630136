"""GPU parity for the code paths no round-1 test executed (VERDICT r1 "What's missing" 4-7, "weak" 2-3): every sub-case goes through
the C-ABI on the HIP path and is compared with the oracle on the same seeded inputs.

  a9   Box / Triangle / Mitchell / Lanczos filter samplers            test_camera_stage_filters_lens_matrix
  a10  thin lens (lens_radius > 0), MatrixCamera record                test_camera_stage_filters_lens_matrix, test_thin_lens_and_spot_frames
  a11  lane_ray_round (closest + any-hit), 32-entry stack              test_lean_traversal_parity
  a14  resolve_mix_material (nested, textured amount), Mix in a frame  test_mix_resolve_bit_exact, test_mix_mirror_frame
  a17  Matte(sigma) / Mirror / Glass / Conductor (RGB, measured)       test_simple_bsdf_pointwise_parity, test_mix_mirror_frame
  a23  Float64 film accumulators                                       test_f64_film_frame
  a24  Point / Spot / Directional / Sun / DiffuseArea sample_light     test_light_sampling_pointwise
  a28  Grid / RGBGrid / NanoVDB sample_point + majorant DDA, bit-exact test_medium_pointwise_bit_exact (incl. the bench-size cloud)
  N4   material_coherence = none / sorted / per_type                   test_material_coherence_is_result_neutral
  configs[3], configs[4] at full size                                  test_full_size_many_light, test_full_size_cloud
"""
import ctypes as C

import numpy as np
import pytest

import ulp_bounds

pytestmark = pytest.mark.gpu
f32 = np.float32


def _pf(hk, a):
    return a.ctypes.data_as(hk._abi.PF)


def _pi(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


def _unit(v):
    return (v / np.linalg.norm(v, axis=1, keepdims=True)).astype(f32)


def ulp_diff(a, b):
    a = np.ascontiguousarray(a, f32).view(np.int32).astype(np.int64)
    b = np.ascontiguousarray(b, f32).view(np.int32).astype(np.int64)
    a = np.where(a < 0, -(a & 0x7fffffff), a)
    b = np.where(b < 0, -(b & 0x7fffffff), b)
    return np.abs(a - b)


def frame_metrics(gpu, ref):
    rel_mse = float(np.mean((gpu - ref) ** 2 / (ref ** 2 + 1e-3)))
    num = np.sqrt(((gpu - ref) ** 2).sum(axis=2))
    den = np.sqrt((ref ** 2).sum(axis=2)) + 1e-6
    return rel_mse, float(np.mean(num / den <= 1e-2))


def _frame_both(hk, oracle, s, cam, w, h, **kw):
    film = hk.Film((w, h))
    vp = hk.VolPath(**kw)
    vp(s, film, cam)
    st = vp.stats()
    vp.close()
    okw = {k: v for k, v in kw.items() if k != "samples_per_pass"}
    acc, ost = oracle.OracleScene(s).render(hk.integrator_params(**okw), cam, w, h, kw["samples"])
    return film.framebuffer.copy(), oracle.finalize(acc, w, h), st, ost


# ---------------------------------------------------------------------------------------------------- a17: simple BSDFs
SIMPLE = ["matte", "matte_sigma", "mirror", "glass", "conductor_rough", "conductor_smooth", "gold_measured"]


@pytest.mark.parametrize("name", SIMPLE)
def test_simple_bsdf_pointwise_parity(hk, oracle, gpu_ctx, name):
    """sample_bsdf / eval_bsdf of the four closed-form kinds (spectral-eval.jl:42-488) on 20 k random (wo, wi, ns, lambda, u, uc),
    random shading normals, with and without regularisation.  Same arithmetic order on both sides: values agree to libm ulps
    (rtol 2e-4 covers sqrt/div-exact code with a few sin/cos), and a Fresnel-vs-uc knife edge may flip a lobe on < 0.05 %."""
    from test_independent_pins import _palette
    s = _palette(hk)
    osc = oracle.OracleScene(s)
    idx = SIMPLE.index(name)
    n = 20000
    rng = np.random.default_rng(17)
    wo, wi, ns = _unit(rng.normal(size=(n, 3))), _unit(rng.normal(size=(n, 3))), _unit(rng.normal(size=(n, 3)))
    lam = (360 + 470 * rng.random((n, 4))).astype(f32)
    u, uc = rng.random((n, 2), dtype=f32), rng.random(n, dtype=f32)
    sh = hk.scene_handle(gpu_ctx, s)
    L = hk._lib.lib()
    for mode in (0, 1):
        for reg in ((False, True) if mode == 0 else (False,)):
            ref = osc.bsdf(mode, idx, wo, wi, ns, lam, u, uc, regularize=reg)
            out = np.zeros((n, 10), f32)
            hk._lib.check(L.hk_test_bsdf(gpu_ctx.h, sh, mode, idx, 1 if reg else 0, n, *[_pf(hk, a) for a in (wo, wi, ns, lam, u, uc, out)]), "hk_test_bsdf")
            assert np.isfinite(out).all()
            close = np.isclose(out, ref, rtol=2e-4, atol=1e-6).all(axis=1)
            assert close.mean() >= 0.9995, (name, mode, reg, close.mean())
            if name in ("matte", "matte_sigma", "mirror"):          # no transcendental beyond sin/cos of the disk map
                assert close.all() and ulp_diff(out[:, 3:8], ref[:, 3:8]).max() <= 4
            # what is achieved on the rows where both sides took the same lobe, against its recorded bound (tests/ulp_bounds.py)
            ulp_bounds.check("bsdf_simple/%s/mode%d/reg%d" % (name, mode, int(reg)), out, ref, floor=1e-3, rows=close)
    osc.close()


# ---------------------------------------------------------------------------------------------------- a14: MixMaterial
def _mix_scene(hk, frame=False, w=48, h=48):
    """Cornell-like box whose sphere carries a NESTED mix (matte | (mirror | conductor)) with a TEXTURED outer amount and whose
    block carries mirror | glass at a constant amount; plus a plain mirror panel."""
    from hikari_jl_amd import geometry as G
    from hikari_jl_amd import scenes
    from hikari_jl_amd.materials import Texture
    rng = np.random.default_rng(23)
    amount = np.clip(rng.random((8, 8)) * 1.4 - 0.2, 0.0, 1.0).astype(f32)      # includes exact 0 and 1 texels (both short-cuts)
    inner = hk.MixMaterial((hk.MirrorMaterial(Kr=hk.RGBSpectrum(0.9, 0.85, 0.7)),
                            hk.ConductorMaterial(eta=hk.RGBSpectrum(0.2, 0.92, 1.1), k=hk.RGBSpectrum(3.9, 2.45, 2.14), roughness=0.05)), 0.35)
    outer = hk.MixMaterial((hk.MatteMaterial(Kd=hk.RGBSpectrum(0.7, 0.3, 0.2)), inner), Texture(amount))
    block = hk.MixMaterial((hk.MirrorMaterial(Kr=hk.RGBSpectrum(0.95)), hk.GlassMaterial(index=1.5)), 0.6)
    white = hk.MatteMaterial(Kd=hk.RGBSpectrum(0.73, 0.73, 0.73))
    s = hk.Scene()
    box, half = 2.0, 1.0
    s.push(G.rect3f((-half, 0, -half), (box, 0.01, box)), white)
    s.push(G.rect3f((-half, box - 0.01, -half), (box, 0.01, box)), white)
    s.push(G.rect3f((-half, 0, half - 0.01), (box, box, 0.01)), white)
    s.push(G.rect3f((-half, 0, -half), (0.01, box, box)), hk.MatteMaterial(Kd=hk.RGBSpectrum(0.65, 0.05, 0.05)))
    s.push(G.rect3f((half - 0.01, 0, -half), (0.01, box, box)), hk.MatteMaterial(Kd=hk.RGBSpectrum(0.12, 0.45, 0.15)))
    s.push(G.sphere((-0.4, 0.4, 0.1), 0.35, 24), outer)
    s.push(G.rect3f((0.15, 0.0, -0.1), (0.5, 0.6, 0.5)), block)
    s.push(G.quad((-0.9, 0.3, 0.6), (-0.3, 0.3, 0.9), (-0.3, 1.4, 0.9), (-0.9, 1.4, 0.6), normal=(0.45, 0.0, -0.89)), hk.MirrorMaterial())
    s.push(hk.PointLight((0, 1.8, 0), hk.RGBSpectrum(15.0)))
    y = 1.98
    s.push(G.quad((-0.25, y, -0.25), (0.25, y, -0.25), (0.25, y, 0.25), (-0.25, y, 0.25), normal=(0, -1, 0)),
           hk.MediumInterface(hk.MatteMaterial(Kd=hk.RGBSpectrum(0.0)), emission=hk.Emissive(Le=hk.RGBSpectrum(1.0), scale=1.0, two_sided=False)))
    s.sync()
    film = hk.Film((w, h))
    cam = hk.PerspectiveCamera((0, 1, -3.5), (0, 1, 0), film, fov=40.0)
    return s, film, cam


def test_mix_resolve_bit_exact(hk, oracle, gpu_ctx):
    """resolve_mix_material (mix-material.jl:96-127, 222-238): 64-bit hash of the bit patterns of (p, wo) and the SetKeys of the two
    children against the nearest-texel amount (Q28); nested mixes are followed.  Integer arithmetic: identical material index
    for every one of 200 k points, for every Mix in the scene."""
    s, _, _ = _mix_scene(hk)
    osc = oracle.OracleScene(s)
    sh = hk.scene_handle(gpu_ctx, s)
    L = hk._lib.lib()
    kinds = [s.desc.materials[i].kind for i in range(s.desc.n_materials)]
    mixes = [i for i, k in enumerate(kinds) if k == hk._abi.HK_MAT_MIX]
    assert len(mixes) == 3
    rng = np.random.default_rng(29)
    n = 200000
    p = (rng.random((n, 3)) * 2 - 1).astype(f32)
    wo = _unit(rng.normal(size=(n, 3)))
    uv = rng.random((n, 2), dtype=f32)
    uv[:100] = np.array([0.0, 0.0], f32)
    uv[100:200] = np.array([1.0, 1.0], f32)
    for m in mixes:
        ref = osc.mix_resolve(m, p, wo, uv)
        out = np.empty(n, np.int32)
        hk._lib.check(L.hk_test_mix(gpu_ctx.h, sh, m, n, _pf(hk, p), _pf(hk, wo), _pf(hk, uv), _pi(out)), "hk_test_mix")
        assert np.array_equal(out, ref), m
        assert all(kinds[i] != hk._abi.HK_MAT_MIX for i in np.unique(out)) and len(np.unique(out)) >= 2
    osc.close()


def test_mix_mirror_frame(hk, oracle):
    """Mix (nested, textured amount) and Mirror objects through the whole K1..K13 loop: k_shade<MIRROR>, resolve_mix_material in
    the trace kernel (queue sorted by the RESOLVED kind), specular chains.  Strict frame tolerance; the ray counts
    of both sides agree because every Mix decision is integer arithmetic on identical bits at the first vertex."""
    w = h = 48
    s, film, cam = _mix_scene(hk, w=w, h=h)
    for depth, spp in ((1, 8), (3, 8), (6, 8)):
        g, r, st, ost = _frame_both(hk, oracle, s, cam, w, h, max_depth=depth, samples=spp)
        rel_mse, frac = frame_metrics(g, r)
        assert np.isfinite(g).all() and rel_mse <= 1e-3 and frac >= 0.99, (depth, rel_mse, frac)
        assert abs(int(st.rays_closest) - int(ost.rays_closest)) <= 0.002 * ost.rays_closest + 4
        assert abs(int(st.rays_shadow) - int(ost.rays_shadow)) <= 0.002 * ost.rays_shadow + 4


# ---------------------------------------------------------------------------------------------------- a24: lights
def _light_scene(hk):
    from hikari_jl_amd import geometry as G
    from hikari_jl_amd.materials import Texture
    rng = np.random.default_rng(31)
    s = hk.Scene()
    s.push(hk.PointLight((0.3, 1.5, 0.2), hk.RGBSpectrum(15.0, 12.0, 9.0)))                                  # scale 1 (Q4)
    s.push(hk.PointLight.from_rgb((2.0, 1.0, 0.5), (-0.4, 0.8, 0.1), power=50.0))                              # illuminant spectrum, power scale
    s.push(hk.SpotLight((0.5, 1.9, -0.5), (0.1, 0, 0.2), hk.RGBSpectrum(30.0), 35.0, 20.0))
    s.push(hk.DirectionalLight(hk.RGBSpectrum(2.0, 1.9, 1.7), (0.2, -1.0, 0.3)))
    s.push(hk.DirectionalLight.from_rgb((3.0, 2.5, 2.0), (0.0, -1.0, 0.0), illuminance=4.0))
    s.push(hk.SunLight.from_rgb((5.0, 4.75, 4.25), (-0.1, -0.2, -0.9)))
    s.push(G.rect3f((-1, 0, -1), (2, 0.01, 2)), hk.MatteMaterial(Kd=Texture(rng.random((4, 4, 3)).astype(f32))))   # gives the scene a texture
    q1 = G.quad((-0.5, 1.2, -0.5), (0.1, 1.2, -0.5), (0.1, 1.2, 0.2), (-0.5, 1.2, 0.2), normal=(0, -1, 0))
    s.push(q1, hk.MediumInterface(hk.MatteMaterial(), emission=hk.Emissive(Le=hk.RGBSpectrum(0.9, 0.6, 0.3), scale=0.8, two_sided=False)))
    q2 = G.quad((0.3, 0.5, 0.3), (0.8, 0.5, 0.3), (0.8, 1.0, 0.5), (0.3, 1.0, 0.5))
    s.push(q2, hk.MediumInterface(hk.MatteMaterial(), emission=hk.Emissive(Le=hk.RGBSpectrum(0.3, 0.5, 0.9), scale=1.0, two_sided=True)))
    s.sync()
    # a TEXTURED area light at the ABI level (the host mirror resolves textured emission per face like scene-mesh.jl:49 does; a
    # `ccall` user may hand the texture over): point the last light's Le at the scene's texture 0
    last = s.desc.n_lights - 1
    assert s.desc.lights[last].kind == hk._abi.HK_LIGHT_DIFFUSE_AREA and s.desc.n_textures >= 1
    s.desc.lights[last].Le.tex = 0
    return s


def test_light_sampling_pointwise(hk, oracle, gpu_ctx):
    """sample_light_spectral for every light kind the point-wise test of round 1 skipped (physical-wavefront/lights.jl:39-131,
    205-297): Point (both constructors), Spot — points inside the inner cone, in the falloff ring and outside the cone —,
    Directional, Sun, DiffuseArea one-sided / two-sided / textured, 100 k points each.  Only +,-,*,/,sqrt on both sides and one
    exp per sigmoid: <= 4 ulp on Li, wi / p_light / pdf exact or <= 2 ulp."""
    s = _light_scene(hk)
    osc = oracle.OracleScene(s)
    sh = hk.scene_handle(gpu_ctx, s)
    L = hk._lib.lib()
    A = hk._abi
    rng = np.random.default_rng(37)
    n = 100000
    p = (rng.random((n, 3)) * np.array([3.0, 2.5, 3.0]) - np.array([1.5, 0.2, 1.5])).astype(f32)
    x = np.zeros((n, 3), f32)
    x[:, :2] = rng.random((n, 2), dtype=f32)
    lam = (360 + 470 * rng.random((n, 4))).astype(f32)
    seen = {}
    for li in range(1, s.desc.n_lights + 1):
        kind = s.desc.lights[li - 1].kind
        ref = osc.light(0, li, p, x, lam)
        out = np.zeros((n, 12), f32)
        hk._lib.check(L.hk_test_light(gpu_ctx.h, sh, 0, li, n, _pf(hk, p), _pf(hk, x), _pf(hk, lam), _pf(hk, out)), "hk_test_light")
        assert np.isfinite(out).all()
        lit = ref[:, 3] > 0
        assert np.array_equal(out[:, 3] > 0, lit), (li, kind)                   # same accept / reject decision at every point
        assert np.array_equal(out[:, 11], ref[:, 11])
        assert ulp_diff(out[lit, 0:3], ref[lit, 0:3]).max() <= 2 and ulp_diff(out[lit, 3], ref[lit, 3]).max() <= 2, (li, kind)
        assert ulp_diff(out[lit, 8:11], ref[lit, 8:11]).max() <= 2
        assert np.allclose(out[lit, 4:8], ref[lit, 4:8], rtol=3e-6, atol=0), (li, kind, np.abs(out[lit, 4:8] / ref[lit, 4:8] - 1).max())
        if kind == A.HK_LIGHT_SPOT:
            l = s.desc.lights[li - 1]
            d = _unit(p - np.array(l.position[:], f32))
            axis = _unit(np.array([[0.1 - 0.5, 0 - 1.9, 0.2 + 0.5]], f32))[0]
            ct = d @ axis
            inner, ring, outside = ct >= l.cos_falloff_start, (ct < l.cos_falloff_start) & (ct >= l.cos_total_width), ct < l.cos_total_width
            assert inner.sum() > 500 and ring.sum() > 500 and outside.sum() > 5000
            assert not lit[outside & (ct < l.cos_total_width - 1e-5)].any() and lit[inner].all()
        if kind == A.HK_LIGHT_DIFFUSE_AREA:
            l = s.desc.lights[li - 1]
            if not l.two_sided:
                assert (~lit).sum() > 1000                                        # points behind a one-sided emitter receive nothing
        seen.setdefault(kind, 0)
        seen[kind] += 1
    assert seen.get(A.HK_LIGHT_POINT) == 2 and seen.get(A.HK_LIGHT_SPOT) == 1 and seen.get(A.HK_LIGHT_DIRECTIONAL) == 2
    assert seen.get(A.HK_LIGHT_SUN) == 1 and seen.get(A.HK_LIGHT_DIFFUSE_AREA, 0) >= 4
    osc.close()


# ---------------------------------------------------------------------------------------------------- a9 / a10: filters, lens, MatrixCamera
def _filters(hk):
    return {"box": hk.BoxFilter(), "triangle": hk.TriangleFilter(), "gaussian": hk.GaussianFilter(), "gaussian_wide": hk.GaussianFilter((2.0, 2.5), 0.8),
            "mitchell": hk.MitchellFilter(), "lanczos": hk.LanczosSincFilter()}


@pytest.mark.parametrize("fname", ["box", "triangle", "gaussian", "gaussian_wide", "mitchell", "lanczos"])
def test_camera_stage_filters_lens_matrix(hk, oracle, gpu_ctx, fname):
    """K1 with every pixel filter (filter.jl:574-604, 733-953: Box and Triangle sample analytically, the others through the
    tabulated 2-D distribution), a thin-lens PerspectiveCamera (perspective.jl:103-112: concentric disk, focal plane) and a
    MatrixCamera record built from Makie-style view / projection matrices (camera/matrix.jl:13-115)."""
    film = hk.Film((200, 120))
    cams = {
        "pinhole": hk.PerspectiveCamera((0, 1, -3.5), (0, 1, 0), film, fov=40.0),
        "thin_lens": hk.PerspectiveCamera((0.5, 1.2, -3.0), (0, 0.8, 0), film, fov=35.0, lens_radius=0.08, focal_distance=3.2),
    }
    from hikari_jl_amd import geometry as G
    view = G.look_at((1.0, 2.0, -4.0), (0, 0.5, 0), (0, 1, 0)).astype(np.float64)
    flip = np.diag([1.0, 1.0, -1.0, 1.0])                       # OpenGL camera looks along -z
    fovy, near, far, aspect = np.deg2rad(45.0), 0.1, 100.0, 200 / 120
    t = 1 / np.tan(fovy / 2)
    proj = np.array([[t / aspect, 0, 0, 0], [0, t, 0, 0], [0, 0, (far + near) / (near - far), 2 * far * near / (near - far)], [0, 0, -1, 0]])
    cams["matrix"] = hk.MatrixCamera(flip @ view, proj, film)
    p = hk.integrator_params(max_depth=5, samples=64, filter=_filters(hk)[fname])
    rng = np.random.default_rng(41)
    n = 20000
    px = rng.integers(1, 201, n).astype(np.int32)
    py = rng.integers(1, 121, n).astype(np.int32)
    si = rng.integers(1, 65, n).astype(np.int32)
    L = hk._lib.lib()
    integ = C.c_void_p()
    hk._lib.check(L.hk_integrator_create(gpu_ctx.h, C.byref(p), C.byref(integ)), "hk_integrator_create")
    for cname, cam in cams.items():
        ref = oracle.camera_samples(p, cam, 200, 120, px, py, si)
        out = np.empty((n, 15), f32)
        rec = cam.record()
        hk._lib.check(L.hk_test_camera(gpu_ctx.h, integ, C.byref(rec), 200, 120, n, _pi(px), _pi(py), _pi(si), _pf(hk, out)), "hk_test_camera")
        assert np.isfinite(out).all(), (fname, cname)
        assert np.allclose(out[:, :4], ref[:, :4], rtol=0, atol=2e-3) and np.allclose(out[:, 4:8], ref[:, 4:8], rtol=2e-5, atol=1e-9)
        assert ulp_diff(out[:, 8], ref[:, 8]).max() <= 2, (fname, cname)                       # filter weight
        if cname == "thin_lens":
            assert ulp_diff(out[:, 9:15], ref[:, 9:15]).max() <= 4, (fname, cname)              # lens point + refocused direction
            assert np.ptp(out[:, 9]) > 0.05                                                     # the origins do spread over the lens
        else:
            assert np.array_equal(out[:, 9:12], ref[:, 9:12]) and ulp_diff(out[:, 12:15], ref[:, 12:15]).max() <= 2, (fname, cname)
    L.hk_integrator_destroy(integ)
    if fname in ("box", "triangle"):
        w = ref[:, 8]
        assert np.allclose(w, w[0])                                                             # analytic samplers: constant weight


def test_thin_lens_and_spot_frames(hk, oracle):
    """Whole frames through k_camera's lens branch (dims 4 and 6 are drawn only when lens_radius > 0) and through a SpotLight
    picked by the light BVH (cone importance, lights.jl:91-98 falloff), each with a non-default filter; strict tolerance."""
    from hikari_jl_amd import scenes
    w = h = 56
    s, film, _ = scenes.cornell_box(w, h, light="area")
    cam = hk.PerspectiveCamera((0, 1, -3.5), (0, 1, 0), film, fov=40.0, lens_radius=0.06, focal_distance=3.4)
    g, r, st, ost = _frame_both(hk, oracle, s, cam, w, h, max_depth=5, samples=8, filter=hk.MitchellFilter())
    rel_mse, frac = frame_metrics(g, r)
    assert rel_mse <= 1e-3 and frac >= 0.99, (rel_mse, frac)
    assert abs(int(st.rays_closest) - int(ost.rays_closest)) <= 0.002 * ost.rays_closest + 4
    pin = hk.PerspectiveCamera((0, 1, -3.5), (0, 1, 0), film, fov=40.0)
    g0, _, _, _ = _frame_both(hk, oracle, s, pin, w, h, max_depth=5, samples=8, filter=hk.MitchellFilter())
    assert np.abs(g - g0).mean() > 1e-3                                                         # and the lens does blur
    from hikari_jl_amd import geometry as G
    s2, film2, cam2 = scenes.cornell_box(w, h, light="none")
    s2.push(hk.SpotLight((0.0, 1.9, -0.3), (0.2, 0.0, 0.2), hk.RGBSpectrum(40.0, 36.0, 30.0), 30.0, 18.0))
    s2.push(hk.SpotLight((-0.8, 1.0, -0.8), (0.4, 0.4, 0.4), hk.RGBSpectrum(10.0, 14.0, 20.0), 25.0, 5.0))
    s2.sync()
    g, r, st, ost = _frame_both(hk, oracle, s2, cam2, w, h, max_depth=4, samples=8, filter=hk.TriangleFilter())
    rel_mse, frac = frame_metrics(g, r)
    assert g.mean() > 1e-3 and rel_mse <= 1e-3 and frac >= 0.99, (rel_mse, frac)
    assert abs(int(st.rays_shadow) - int(ost.rays_shadow)) <= 0.002 * ost.rays_shadow + 4


# ---------------------------------------------------------------------------------------------------- a23: Float64 film
def test_f64_film_frame(hk, oracle):
    """accumulation_eltype = Float64 (volpath.jl:38, 83, 86): k_film<double> / k_finalize<double>.  Against the oracle's f64
    accumulators (strict frame tolerance); against the f32 film of the same samples (identical per-sample values, different sum
    precision: <= 1e-6 relative, and NOT bit-equal over 64 samples, which shows the double path really ran); progressive adds
    on top of f64 accumulators are bit-identical to one call."""
    from hikari_jl_amd import scenes
    w = h = 40
    s, film, cam = scenes.cornell_box(w, h, light="area")
    kw = dict(max_depth=5, samples=64)
    g64, r64, st, ost = _frame_both(hk, oracle, s, cam, w, h, accumulation_eltype="Float64", **kw)
    rel_mse, frac = frame_metrics(g64, r64)
    assert rel_mse <= 1e-3 and frac >= 0.99
    f32film = hk.Film((w, h))
    v32 = hk.VolPath(**kw)
    v32(s, f32film, cam)
    a32 = v32.read_accumulators(f32film)
    v32.close()
    v64 = hk.VolPath(accumulation_eltype="Float64", **kw)
    f64film = hk.Film((w, h))
    v64(s, f64film, cam)
    a64 = v64.read_accumulators(f64film)
    assert a64.dtype == np.float64 and a32.dtype == np.float32
    assert np.allclose(a64, a32, rtol=2e-6, atol=1e-7) and not np.array_equal(a64.astype(np.float32), a32)
    # the f64 sum of f32 terms is exact to ~1e-16 relative: re-rendering in 8 progressive steps gives the same doubles
    v64.clear()
    f64film.iteration_index = 0
    for k in range(8):
        v64.render_samples(s, f64film, cam, 8, readback=False)
    assert np.array_equal(v64.read_accumulators(f64film), a64)
    v64.close()
    # the ABI rejects a film / integrator precision mismatch instead of reinterpreting memory
    L = hk._lib.lib()
    ctx = hk.Context.get(0)
    p64 = hk.integrator_params(accumulation_eltype="Float64", **kw)
    integ, fh = C.c_void_p(), C.c_void_p()
    hk._lib.check(L.hk_integrator_create(ctx.h, C.byref(p64), C.byref(integ)), "hk_integrator_create")
    hk._lib.check(L.hk_film_create(ctx.h, w, h, 0, None, C.byref(fh)), "hk_film_create")
    rec = cam.record()
    assert L.hk_render(ctx.h, hk.scene_handle(ctx, s), integ, fh, C.byref(rec), 1, 1, 1) == hk._abi.HK_ERR_INVALID
    L.hk_film_destroy(fh)
    L.hk_integrator_destroy(integ)


# ---------------------------------------------------------------------------------------------------- a28: media, point-wise
def _media_scene(hk, which):
    """-> (scene, medium index, bounding box lo / hi in render space)"""
    from hikari_jl_amd import scenes
    rng = np.random.default_rng(43)
    if which == "homogeneous":
        med = hk.HomogeneousMedium(sigma_a=hk.RGBSpectrum(0.2, 0.3, 0.1), sigma_s=hk.RGBSpectrum(0.8, 0.6, 0.9), Le=hk.RGBSpectrum(0.05, 0.0, 0.02), g=0.4)
        s, _, _ = scenes.slab_scene(16, 16, med)
        return s, (-2.5, -2.6, 1.0), (2.5, 2.6, 2.0)
    if which == "grid":
        dens = (rng.random((21, 17, 9)) ** 2).astype(f32) * 3.0
        med = hk.GridMedium(dens, sigma_a=hk.RGBSpectrum(0.1, 0.2, 0.3), sigma_s=hk.RGBSpectrum(1.0, 0.9, 0.8), g=-0.2, bounds=((-2.5, -2.6, 1.0), (2.5, 2.6, 2.0)),
                            majorant_res=(5, 4, 3))
        s, _, _ = scenes.slab_scene(16, 16, med)
        return s, (-2.5, -2.6, 1.0), (2.5, 2.6, 2.0)
    if which == "rgbgrid":
        sa = (rng.random((10, 8, 6, 3)) * np.array([0.6, 0.9, 1.3])).astype(f32)
        ss = (rng.random((10, 8, 6, 3)) ** 2 * np.array([2.0, 1.6, 1.2])).astype(f32)
        le = (rng.random((10, 8, 6, 3)) * np.array([0.3, 0.1, 0.02])).astype(f32)
        med = hk.RGBGridMedium(sigma_a_grid=sa, sigma_s_grid=ss, Le_grid=le, sigma_scale=1.5, Le_scale=0.8, g=0.3, bounds=((-2.5, -2.6, 1.0), (2.5, 2.6, 2.0)),
                               majorant_res=(4, 4, 2))
        s, _, _ = scenes.slab_scene(16, 16, med)
        return s, (-2.5, -2.6, 1.0), (2.5, 2.6, 2.0)
    if which == "nanovdb_small":
        s, _, _ = scenes.cloud_scene(16, 16, "nanovdb", res=(48, 48, 24))
        return s, (-0.6, 0.3, -0.6), (0.6, 1.5, 0.6)
    assert which == "nanovdb_bench"            # BASELINE configs[3] exactly as bench.py builds it (5 % fill, extinction 620, 64^3 majorant)
    s, _, _ = scenes.bomex_scene(16, 16)
    return s, (-0.6, 0.3, -0.6), (0.6, 1.5, 0.6)


@pytest.mark.parametrize("which", ["homogeneous", "grid", "rgbgrid", "nanovdb_small", "nanovdb_bench"])
def test_medium_pointwise_bit_exact(hk, oracle, gpu_ctx, which):
    """sample_point and the majorant DDA of every medium kind, device vs oracle, 100 k points / 50 k rays each, BIT-EXACT.
    For NanoVDB the two sides are different algorithms: the oracle walks root -> upper -> lower -> leaf per voxel tap like the
    reference (nanovdb.jl:315-388), the device reads the host-flattened block table (hk_nanovdb.h) — values and the trilinear
    blend (:400-469) must be identical, inside, on the boundary of and outside the grid's index bounding box.  RGBGrid applies a
    per-point sigmoid uplift (one exp per wavelength): <= 2 ulp there, exact elsewhere."""
    s, lo, hi = _media_scene(hk, which)
    osc = oracle.OracleScene(s)
    sh = hk.scene_handle(gpu_ctx, s)
    L = hk._lib.lib()
    lo, hi = np.array(lo, f32), np.array(hi, f32)
    rng = np.random.default_rng(47)
    n = 100000
    p = (lo - 0.1 * (hi - lo) + 1.2 * (hi - lo) * rng.random((n, 3))).astype(f32)              # 20 % of the points fall outside the bounds
    p[:300] = lo
    p[300:600] = hi
    p[600:900, 0] = lo[0]
    lam = (360 + 470 * rng.random((n, 4))).astype(f32)
    ref = osc.medium(0, 0, p, lam)
    out = np.zeros((n, 13), f32)
    hk._lib.check(L.hk_test_medium(gpu_ctx.h, sh, 0, 0, n, _pf(hk, p), None, None, _pf(hk, lam), _pf(hk, out)), "hk_test_medium")
    assert np.isfinite(out).all() and (ref[:, 4:8] > 0).mean() > 0.02
    if which == "rgbgrid":
        assert ulp_diff(out, ref).max() <= 2
    else:
        assert np.array_equal(out, ref), (which, np.abs(out - ref).max())
    m = 50000
    o = (lo - 0.5 * (hi - lo) + 2.0 * (hi - lo) * rng.random((m, 3))).astype(f32)
    target = (lo + (hi - lo) * rng.random((m, 3))).astype(f32)
    d = _unit(target - o)
    d[:200] = np.array([0, 0, 1], f32)                                                          # axis-parallel rays (infinite DDA steps on two axes)
    d[200:400] = np.array([1, 0, 0], f32)
    tmax = np.where(rng.random(m) < 0.3, rng.random(m) * 3.0, np.inf).astype(f32)
    refm = osc.medium(1, 0, o, lam[:m], d, tmax)
    outm = np.zeros((m, 49), f32)
    hk._lib.check(L.hk_test_medium(gpu_ctx.h, sh, 1, 0, m, _pf(hk, o), _pf(hk, d), _pf(hk, tmax), _pf(hk, lam[:m]), _pf(hk, outm)), "hk_test_medium")
    assert np.array_equal(outm[:, 0], refm[:, 0]) and refm[:, 0].max() >= (1 if which == "homogeneous" else 4)
    assert np.array_equal(outm, refm), (which, np.abs(outm - refm).max())
    # mode 2: the walk the tracking kernels really do — cells whose majorant is exactly 0 are fast-forwarded without fetching the
    # grid (majorant_skip_zero).  Same total segment count (the 256-segment cap counts skipped cells too) and, among the oracle's
    # first 16 segments, exactly the non-zero ones, bit for bit, in order.
    out2 = np.zeros((m, 49), f32)
    hk._lib.check(L.hk_test_medium(gpu_ctx.h, sh, 2, 0, m, _pf(hk, o), _pf(hk, d), _pf(hk, tmax), _pf(hk, lam[:m]), _pf(hk, out2)), "hk_test_medium")
    assert np.array_equal(out2[:, 0], refm[:, 0])
    segs_ref, segs_gpu = refm[:, 1:].reshape(m, 16, 3), out2[:, 1:].reshape(m, 16, 3)
    n_ref = np.minimum(refm[:, 0], 16).astype(int)
    skipped_any = 0
    for i in range(0, m, 7):
        keep = [k for k in range(n_ref[i]) if segs_ref[i, k, 2] != 0.0]
        skipped_any += len(keep) < n_ref[i]
        assert np.array_equal(segs_gpu[i, :len(keep)], segs_ref[i, keep]), (which, i)
    if which not in ("homogeneous", "rgbgrid", "grid"):
        assert skipped_any > 100                                                                # the clouds do have empty cells
    osc.close()


# ---------------------------------------------------------------------------------------------------- a11: the lean traversal
@pytest.mark.parametrize("which", ["cornell", "cornell_two_spheres", "many_light_full"])
def test_lean_traversal_parity(hk, oracle, gpu_ctx, which):
    """The traversal the bench path runs — lane_ray_round (while-while rounds, straggler exit, per-lane refill) with the LDS stack
    the BVH depth selects — against the oracle's closest hit: (t, prim, bary) bit-exact on 1 M rays; any-hit mode: occluded
    <=> the oracle's closest hit exists inside t_max, and the reported hit is a genuine one (its t equals the oracle's t for
    that primitive or lies behind the closest one).  `many_light_full` is the 10^6-triangle scene of BASELINE configs[4]: the only
    BVH deeper than 16 levels, i.e. the only user of the 32-entry-stack instantiations."""
    from hikari_jl_amd import scenes
    if which.startswith("cornell"):
        # `cornell_two_spheres` is SURVEY 8(d)'s own geometry (3 782 triangles): its BVH does not fit the 1 536-node LDS cache whole, so
        # one traversal mixes cached and global nodes
        s, _, _ = scenes.cornell_box(64, 64) if which == "cornell" else scenes.cornell_box(64, 64, light="area", objects="two_spheres")
        lo, hi = np.array([-1.2, -0.2, -1.2]), np.array([1.2, 2.2, 1.2])
    else:
        s, _, _ = scenes.many_light_scene(64, 64)
        lo, hi = np.array([-7.0, -7.0, -7.0]), np.array([7.0, 7.0, 7.0])
    sh = hk.scene_handle(gpu_ctx, s)
    L = hk._lib.lib()
    depth = C.c_int32()
    L.hk_scene_bvh_info(sh, None, None, C.byref(depth))
    assert (depth.value > 16) == (which == "many_light_full"), depth.value
    rng = np.random.default_rng(53)
    n = 1_000_000
    o = (lo + (hi - lo) * rng.random((n, 3))).astype(f32)
    d = _unit(rng.normal(size=(n, 3)))
    d[:1000] = np.array([0, -1, 0], f32)
    d[1000:2000] = np.array([1, 0, 0], f32)
    tmax = np.full(n, np.inf, f32)
    tmax[::5] = (rng.random(len(tmax[::5])) * (2.0 if which.startswith("cornell") else 6.0)).astype(f32)
    tmax[7::97] = 0.0
    osc = oracle.OracleScene(s)
    rt, rp, ruv = osc.trace(o, d, tmax)
    assert 0.2 < (rp >= 0).mean() < 0.999
    args = [_pf(hk, o), _pf(hk, d), _pf(hk, tmax)]
    gt, gp, guv = np.empty(n, f32), np.empty(n, np.int32), np.empty((n, 2), f32)
    hk._lib.check(L.hk_test_trace_lean(gpu_ctx.h, sh, 0, n, *args, _pf(hk, gt), _pi(gp), _pf(hk, guv)), "hk_test_trace_lean")
    assert np.array_equal(gp, rp) and np.array_equal(gt, rt) and np.array_equal(guv, ruv)
    # the general-path traversal (traverse<>) on the same rays: the deep scene was never run through it either
    hk._lib.check(L.hk_trace_closest(gpu_ctx.h, sh, n, *args, _pf(hk, gt), _pi(gp), _pf(hk, guv)), "hk_trace_closest")
    assert np.array_equal(gp, rp) and np.array_equal(gt, rt) and np.array_equal(guv, ruv)
    at, ap, auv = np.empty(n, f32), np.empty(n, np.int32), np.empty((n, 2), f32)
    hk._lib.check(L.hk_test_trace_lean(gpu_ctx.h, sh, 1, n, *args, _pf(hk, at), _pi(ap), _pf(hk, auv)), "hk_test_trace_lean")
    assert np.array_equal(ap >= 0, rp >= 0)                                                      # occlusion decision identical
    occ = ap >= 0
    assert (at[occ] >= rt[occ]).all() and (at[occ] < tmax[occ]).all()                            # a real hit, never in front of the closest one
    same = occ & (ap == rp)
    assert same.mean() > 0.2 and np.array_equal(at[same], rt[same])
    osc.close()


# ---------------------------------------------------------------------------------------------------- N4: material_coherence
def test_material_coherence_is_result_neutral(hk):
    """material_coherence = :none / :sorted / :per_type (multi-material-eval.jl:169-286, 516-549) only reorders material
    evaluation in the reference; here the queue is always compacted per resolved kind.  All three values are accepted and give
    bit-identical films and ray counts on a scene with seven material kinds incl. Mix."""
    w = h = 40
    s, film, cam = _mix_scene(hk, w=w, h=h)
    frames, rays = [], []
    for mode in ("none", "sorted", "per_type"):
        f = hk.Film((w, h))
        vp = hk.VolPath(max_depth=5, samples=4, material_coherence=mode)
        assert vp.params.material_coherence == ("none", "sorted", "per_type").index(mode)
        vp(s, f, cam)
        frames.append(f.framebuffer.copy())
        st = vp.stats()
        rays.append((int(st.rays_closest), int(st.rays_shadow)))
        vp.close()
    assert np.array_equal(frames[0], frames[1]) and np.array_equal(frames[0], frames[2]) and rays[0] == rays[1] == rays[2]
    assert frames[0].mean() > 0.01


# ---------------------------------------------------------------------------------------------------- configs[3] / configs[4] at full size
def test_full_size_many_light(hk, oracle):
    """BASELINE configs[4] stand-in at FULL size (10^6 triangles, ~5 * 10^4 area lights, 1024^2, depth 8), one sample per pixel:
    size-independent properties — finite, non-negative, deterministic, sharded == whole (sample-index sharding over 2 ranks
    bit-equal in the accumulators up to fp32 summation order), filter weight sum constant — and oracle parity on a 96 x 96 crop
    rendered through the SAME scene (the oracle builds its own BVH over the 10^6 triangles): strict frame tolerance."""
    from hikari_jl_amd import scenes
    s, film, cam = scenes.many_light_scene(1024, 1024)
    assert s.desc.n_triangles >= 1_000_000 and s.desc.n_lights > 40_000
    vp = hk.VolPath(max_depth=8, samples=2)
    vp(s, film, cam)
    a = film.framebuffer.copy()
    acc = vp.read_accumulators(film)
    st = vp.stats()
    assert np.isfinite(a).all() and (a >= 0).all() and a.mean() > 1e-3
    assert st.rays_closest >= 2 * 1024 * 1024 and st.rays_shadow > 1024 * 1024
    wsum = acc[3 * 1024 * 1024:]
    assert np.allclose(wsum, wsum[0], rtol=1e-2)
    vp(s, film, cam)
    assert np.array_equal(a, film.framebuffer)
    parts = []
    for rank in range(2):
        f = hk.Film((1024, 1024))
        v = hk.VolPath(max_depth=8, samples=2)
        v._ensure(f)
        v.clear()
        v.render_samples(s, f, cam, 1, stride=2, first=rank + 1, readback=False)
        parts.append(v.read_accumulators(f))
        v.close()
    assert np.allclose(parts[0] + parts[1], acc, rtol=1e-5, atol=1e-6)
    vp.close()
    w = h = 96
    small = hk.Film((w, h))
    cam_s = hk.PerspectiveCamera((0.0, -0.2, 9.0), (0.8, 0.3, 0.0), small, up=(0, 1, 0), fov=60.0)
    g, r, st, ost = _frame_both(hk, oracle, s, cam_s, w, h, max_depth=4, samples=2)
    rel_mse, frac = frame_metrics(g, r)
    assert rel_mse <= 1e-3 and frac >= 0.99, (rel_mse, frac)
    assert abs(int(st.rays_closest) - int(ost.rays_closest)) <= 0.002 * ost.rays_closest + 4


def test_collision_bytes_priced_per_scene(hk):
    """hk_stats' algorithmic bytes charge 36 B per collision in a dense grid and 84 B through a NanoVDB tree.  One
    integrator — one statistics window — that renders a grid scene and then a NanoVDB scene must charge each scene's collisions
    their own price: the bytes add up to those of the two scenes rendered alone (a per-context "a NanoVDB scene was seen" flag
    would charge the grid's collisions 84 B as well)."""
    from hikari_jl_amd import scenes
    kw = dict(res=(32, 32, 16), sigma_scale=20.0, majorant_res=(8, 8, 8))
    sg, fg, cg = scenes.cloud_scene(48, 40, kind="grid", **kw)
    sn, fn, cn = scenes.cloud_scene(48, 40, kind="nanovdb", **kw)
    vp = hk.VolPath(max_depth=6, samples=2)
    vp(sg, fg, cg)
    g = vp.stats()
    bg, bsg, cgl = int(g.bytes_algorithmic_media), int(g.bytes_algorithmic_shadow), int(g.track_collisions)
    assert cgl > 1000 and int(g.shadow_collisions) > 100
    vp.reset_stats()
    vp(sn, fn, cn)
    n = vp.stats()
    bn, bsn = int(n.bytes_algorithmic_media), int(n.bytes_algorithmic_shadow)
    assert int(n.track_collisions) > 1000
    assert bn >= 84 * int(n.track_collisions) and bsn >= 84 * int(n.shadow_collisions)
    vp.reset_stats()                                      # (VolPath.__call__ resets the statistics: render_samples does not)
    fg.iteration_index = fn.iteration_index = 0           # (the same two samples as above)
    vp.clear()
    vp.render_samples(sg, fg, cg, 2)
    vp.clear()
    vp.render_samples(sn, fn, cn, 2)
    both = vp.stats()
    assert int(both.bytes_algorithmic_media) == bg + bn and int(both.bytes_algorithmic_shadow) == bsg + bsn
    vp.close()


def test_full_size_cloud(hk, oracle):
    """BASELINE configs[3] at FULL size exactly as bench.py builds it (`scenes.bomex_scene`: 256 x 256 x 128 worley-fbm field at 5 %
    fill, extinction up to 620, NanoVDB + 64^3 majorant, 1024^2, depth 32), one sample per pixel: finite, non-negative,
    deterministic (film and collision counters), sample-sharded == unsharded.  Plus the ABSORBING-only variant of the same field — no
    scattering, so no re-seeding from direction bits — against the oracle on a 64 x 64 frame: strict tolerance and
    an identical collision count (delta tracking consumes the same RNG stream on both sides)."""
    from hikari_jl_amd import scenes
    s, film, cam = scenes.bomex_scene(1024, 1024)
    vp = hk.VolPath(max_depth=32, samples=2)
    vp(s, film, cam)
    a = film.framebuffer.copy()
    acc = vp.read_accumulators(film)
    st = vp.stats()
    c1 = int(st.medium_collisions)
    assert np.isfinite(a).all() and (a >= 0).all() and a.mean() > 0.01 and c1 > 1_000_000
    assert int(st.track_collisions) + int(st.shadow_collisions) == c1 and int(st.track_dda_steps) > 0 and int(st.shadow_dda_steps) > 0
    assert int(st.scatter_vertices) > 100_000 and int(st.bytes_algorithmic_media) >= 84 * int(st.track_collisions)
    vp(s, film, cam)
    assert np.array_equal(a, film.framebuffer) and int(vp.stats().medium_collisions) == c1
    # the two samples rendered as two strided calls (two ranks' shares) add up to the same film, up to the order of the two fp32 adds
    vp.clear()
    vp.render_samples(s, film, cam, 1, stride=2, first=1, readback=False)
    vp.render_samples(s, film, cam, 1, stride=2, first=2, readback=False)
    acc2 = vp.read_accumulators(film)
    assert np.allclose(acc, acc2, rtol=1e-6, atol=1e-7)
    vp.close()
    s2, _, _ = scenes.bomex_scene(64, 64, max_extinction=620.0 / 16, sigma_a=hk.RGBSpectrum(0.9, 1.0, 1.2), sigma_s=hk.RGBSpectrum(0.0), g=0.0)
    w = h = 64
    f2 = hk.Film((w, h))
    cam2 = hk.PerspectiveCamera((0.0, 1.0, -3.5), (0.0, 0.9, 0.0), f2, fov=40.0)
    g, r, st, ost = _frame_both(hk, oracle, s2, cam2, w, h, max_depth=6, samples=8)
    rel_mse, frac = frame_metrics(g, r)
    assert rel_mse <= 1e-3 and frac >= 0.99, (rel_mse, frac)
    assert int(st.medium_collisions) == int(ost.medium_collisions) and int(st.medium_collisions) > 10_000


# ---------------------------------------------------------------------------------------------------- a8: tabulated ZSobol digits
@pytest.mark.parametrize("spp_setting", [64, 8192, 40000])
def test_zsobol_pixel_table_all_index_widths(hk, oracle, spp_setting):
    """The render kernels read the permuted pixel digits and the permutation indices of the two top sample digits from a table
    (k_sobol_table, zsobol_top_perms); hk_test_sobol exercises the untabulated arithmetic only.  Frames are therefore compared
    with the oracle for an even (log2 = 12: every `samples` <= 4096), an odd (8192 -> 13) and a wide (40000 -> 16) index width —
    a handful of samples of each, on a ragged film — bit-tight: a wrong digit would decorrelate every pixel."""
    from hikari_jl_amd import scenes
    w, h = 37, 29
    s, film, cam = scenes.cornell_box(w, h, light="area")
    kw = dict(max_depth=4, samples=spp_setting)
    p = hk.integrator_params(**kw)
    osc = oracle.OracleScene(s)
    acc, ost = osc.render(p, cam, w, h, 3, first=5)
    ref = oracle.finalize(acc, w, h)
    vp = hk.VolPath(**kw)
    vp._ensure(film)
    vp.clear()
    vp.reset_stats()
    vp.render_samples(s, film, cam, 3, first=5)
    st = vp.stats()
    vp.close()
    rel_mse, frac = frame_metrics(film.framebuffer, ref)
    assert rel_mse <= 1e-5 and frac >= 0.995, (spp_setting, rel_mse, frac)
    assert abs(int(st.rays_closest) - int(ost.rays_closest)) <= 0.002 * ost.rays_closest + 4


@pytest.mark.parametrize("first,n,stride,per_pass,spp_setting", [(1, 16, 1, 0, 64), (3, 21, 1, 8, 64), (2, 17, 3, 5, 64), (1, 20, 1, 0, 8192), (6, 16, 2, 0, 40000)])
def test_zsobol_sample_bit_table(hk, oracle, knobs, first, n, stride, per_pass, spp_setting):
    """A call that renders >= 16 samples reads the permuted sample bits from a table (k_sobol_lo_table, DSobol::lo_table) instead of
    hashing the remaining base-4 digits.  The table must not change a single bit: the accumulators of the same call with the table
    switched off (HK_SOBOL_LO_GB=0) are compared exactly — over aligned and unaligned first samples, strides, several passes per
    call (per-pass table offset), the last pass shorter than the others, even / odd / wide index widths — and the frame with the oracle."""
    from hikari_jl_amd import scenes
    w, h = 21, 19
    s, film, cam = scenes.cornell_box(w, h, light="area")
    kw = dict(max_depth=4, samples=spp_setting, samples_per_pass=per_pass)

    def run():
        vp = hk.VolPath(**kw)
        vp._ensure(film)
        vp.clear()
        vp.render_samples(s, film, cam, n, stride=stride, first=first)
        acc = vp.read_accumulators(film).copy()
        fb = film.framebuffer.copy()
        vp.close()
        return acc, fb

    acc_table, fb = run()
    knobs.setenv("HK_SOBOL_LO_GB", "0.0002")     # 200 kB: room for a handful of the 29 rows — the deeper dimensions hash their digits
    acc_partial, _ = run()
    knobs.setenv("HK_SOBOL_LO_GB", "0")
    acc_hashed, _ = run()
    assert np.array_equal(acc_table.view(np.uint32), acc_hashed.view(np.uint32))
    assert np.array_equal(acc_partial.view(np.uint32), acc_hashed.view(np.uint32))
    osc = oracle.OracleScene(s)
    p = hk.integrator_params(max_depth=4, samples=spp_setting)
    oacc, _ = osc.render(p, cam, w, h, n, first=first, stride=stride)
    rel_mse, frac = frame_metrics(fb, oracle.finalize(oacc, w, h))
    assert rel_mse <= 1e-5 and frac >= 0.995, (rel_mse, frac)


def test_light_preselection_is_result_neutral(hk, knobs):
    """Scenes with a deep light BVH choose the next-event light in a kernel of its own (k_light_select: per-lane descent with refill)
    before the shade kernels run.  Same arithmetic per vertex as the fused form: the film must be bit-identical with HK_PRESELECT=0,
    through the pooled kernel (k_light_select_pool, the default since round 5) and the per-lane-refill one (HK_SELECT_POOL=0) at any
    refill threshold, with static and ticketed segments, with and without the table-only Sobol instantiations."""
    from hikari_jl_amd import scenes
    w, h = 40, 36
    s, film, cam = scenes.many_light_scene(w, h, n_boxes=4000)      # ~2.4 k area lights: well above HK_PRESELECT_MIN
    kw = dict(max_depth=5, samples=64)

    names = ("HK_PRESELECT", "HK_SELECT_MIN_IDLE", "HK_SOBOL_TABLE_ONLY", "HK_SELECT_POOL", "HK_DYNAMIC_SEGMENTS", "HK_WAVES_PER_CU", "HK_SMALL_PASS")

    def run(env):
        for k in names:
            knobs.delenv(k, raising=False)
        for k, v in env.items():
            knobs.setenv(k, v)
        vp = hk.VolPath(**kw)
        vp._ensure(film)
        vp.clear()
        vp.reset_stats()
        vp.render_samples(s, film, cam, 20, first=1)
        acc = vp.read_accumulators(film).copy()
        nodes = int(vp.stats().light_bvh_nodes)
        vp.close()
        return acc, nodes

    ref, nodes = run({"HK_PRESELECT": "0"})
    assert np.isfinite(ref).all() and ref.max() > 0 and nodes > 0
    old = {"HK_SELECT_POOL": "0"}
    for env in ({}, {"HK_SOBOL_TABLE_ONLY": "0"}, {"HK_DYNAMIC_SEGMENTS": "1"}, {"HK_DYNAMIC_SEGMENTS": "0", "HK_WAVES_PER_CU": "5"}, {"HK_DYNAMIC_SEGMENTS": "1", "HK_WAVES_PER_CU": "3"},
                {"HK_SMALL_PASS": "0"}, old, {**old, "HK_SELECT_MIN_IDLE": "1"}, {**old, "HK_SELECT_MIN_IDLE": "64"}, {**old, "HK_SOBOL_TABLE_ONLY": "0"},
                {"HK_SOBOL_TABLE_ONLY": "0", "HK_PRESELECT": "0"}):
        got, n2 = run(env)
        assert np.array_equal(ref.view(np.uint32), got.view(np.uint32)), env
        assert n2 == nodes, (env, n2, nodes)                            # the same node evaluations, wherever they run
    for k in names:
        knobs.delenv(k, raising=False)


def test_quantised_nodes_are_result_neutral(hk, knobs, gpu_ctx):
    """Trees deeper than 16 levels are traversed through DQNode (16-bit planes on a grid over the scene's bounds, boxes only ever larger
    than the float ones: hk_types.h) by the lean kernels.  A scene created with HK_QNODES=0 keeps the float nodes: films (accumulators,
    bit for bit), ray counts and a 300 k-ray closest-hit / any-hit table are the same — what the boxes prune differs, what the triangle
    tests find does not (ties on t go to the smaller triangle index, whatever the visiting order)."""
    from hikari_jl_amd import scenes
    L = hk._lib.lib()
    w, h = 48, 40
    kw = dict(max_depth=5, samples=8)

    def run(env):
        knobs.delenv("HK_QNODES", raising=False)
        knobs.setenv("HK_SMALL_PASS_FUSED", "0")                                     # (a film this small would be ONE k_small_pass launch, which reads the float nodes: the stages as launches)
        for k, v in env.items():
            knobs.setenv(k, v)
        s, film, cam = scenes.many_light_scene(w, h, n_boxes=12000, seed=5, box_scale=2.5)        # 144 k triangles; a fresh Scene object: the knob is read when its device scene is built
        sh = hk.scene_handle(gpu_ctx, s)
        depth = C.c_int32()
        L.hk_scene_bvh_info(sh, None, None, C.byref(depth))
        assert depth.value > 16, depth.value
        vp = hk.VolPath(**kw)
        vp._ensure(film)
        vp.clear()
        vp.reset_stats()
        vp.render_samples(s, film, cam, 8, first=1)
        acc = vp.read_accumulators(film).copy()
        st = vp.stats()
        assert int(st.fused_passes) == 0
        rays = (int(st.rays_closest), int(st.rays_shadow), int(st.path_vertices))
        vp.close()
        rng = np.random.default_rng(11)
        n = 300_000
        o = (rng.random((n, 3)) * 14 - 7).astype(f32)
        d = _unit(rng.normal(size=(n, 3)))
        d[:500] = np.array([0, 0, 1], f32)
        tmax = np.full(n, np.inf, f32)
        tmax[::3] = (rng.random(len(tmax[::3])) * 6.0).astype(f32)
        out = []
        for anyhit in (0, 1):
            t, pr, uv = np.empty(n, f32), np.empty(n, np.int32), np.empty((n, 2), f32)
            hk._lib.check(L.hk_test_trace_lean(gpu_ctx.h, sh, anyhit, n, _pf(hk, o), _pf(hk, d), _pf(hk, tmax), _pf(hk, t), _pi(pr), _pf(hk, uv)), "hk_test_trace_lean")
            out.append((t, pr, uv))
        return acc, rays, out

    ref, rays0, tab0 = run({"HK_QNODES": "0"})
    got, rays1, tab1 = run({})
    knobs.delenv("HK_QNODES", raising=False)
    knobs.delenv("HK_SMALL_PASS_FUSED", raising=False)
    assert np.isfinite(ref).all() and ref.max() > 0
    assert np.array_equal(ref.view(np.uint32), got.view(np.uint32)) and rays0 == rays1
    (t0, p0, u0), (t1, p1, u1) = tab0[0], tab1[0]
    assert np.array_equal(p0, p1) and np.array_equal(t0.view(np.uint32), t1.view(np.uint32)) and np.array_equal(u0.view(np.uint32), u1.view(np.uint32))
    assert 0.01 < (p0 >= 0).mean() < 0.999, (p0 >= 0).mean()
    assert np.array_equal(tab0[1][1] >= 0, tab1[1][1] >= 0)                         # any hit: the same rays are occluded


@pytest.mark.parametrize("which", ["cornell", "sky", "slab", "cloud", "cloud_grid"])
def test_scheduling_is_result_neutral(hk, knobs, which):
    """How segments reach waves must not change a bit of the film: static stride vs tickets over the work lists, other segment counts
    (a count that is no multiple of anything), the shadow kernels on the second stream or not, BVH nodes from LDS or from global memory.
    Accumulators compared exactly."""
    from hikari_jl_amd import scenes
    w, h = 40, 36
    if which == "cornell":
        s, film, cam = scenes.cornell_box(w, h, light="area")
        kw = dict(max_depth=6, samples=64)
    elif which == "sky":
        s, film, cam = scenes.sky_scene(w, h, env_res=32)
        kw = dict(max_depth=6, samples=64)
    elif which == "cloud":   # a GREY NanoVDB medium (flat sigma_a / sigma_s): the specialised tracking kernels and their loop-shape knobs
        s, film, cam = scenes.cloud_scene(w, h, "nanovdb", res=(48, 48, 24))
        kw = dict(max_depth=8, samples=64)
    elif which == "cloud_grid":   # the same field as a dense GridMedium: the pool kernels without the NanoVDB bricks
        s, film, cam = scenes.cloud_scene(w, h, "grid", res=(48, 48, 24))
        kw = dict(max_depth=8, samples=64)
    else:
        s, film, cam = scenes.slab_scene(w, h, hk.HomogeneousMedium(sigma_a=hk.RGBSpectrum(0.2), sigma_s=hk.RGBSpectrum(0.8, 0.7, 0.6), g=0.3))
        kw = dict(max_depth=6, samples=64)
    names = ("HK_OVERLAP", "HK_DYNAMIC_SEGMENTS", "HK_WAVES_PER_CU", "HK_NODE_CACHE", "HK_WALK_SPLIT", "HK_GREY", "HK_DELTA_ADVANCE", "HK_TRACK_ADVANCE",
             "HK_SHADOW_TRACK_BATCH", "HK_SHADOW_FEED_ROUNDS", "HK_TRACK_REFILL_IDLE", "HK_WALK_REFILL_IDLE", "HK_GREY_FLAT", "HK_TRACK_POOL", "HK_WALK_POOL", "HK_SMALL_PASS", "HK_SMALL_PASS_WAVES", "HK_TICKET_SHARE", "HK_GREY_COMPACT", "HK_SMALL_PASS_FUSED", "HK_SHADOW_FINAL", "HK_ESCAPED_UNROLL", "HK_LEAN_RECORDS")

    def run(env):
        for k in names:
            knobs.delenv(k, raising=False)
        for k, v in env.items():
            knobs.setenv(k, v)
        vp = hk.VolPath(**kw)
        vp._ensure(film)
        vp.clear()
        vp.render_samples(s, film, cam, 20, first=1)
        acc = vp.read_accumulators(film).copy()
        vp.close()
        return acc

    ref = run({})
    assert np.isfinite(ref).all() and ref.max() > 0
    for env in ({"HK_OVERLAP": "0"}, {"HK_DYNAMIC_SEGMENTS": "0"}, {"HK_DYNAMIC_SEGMENTS": "1"}, {"HK_DYNAMIC_SEGMENTS": "1", "HK_WAVES_PER_CU": "7"},
                {"HK_DYNAMIC_SEGMENTS": "0", "HK_WAVES_PER_CU": "5", "HK_OVERLAP": "1"}, {"HK_NODE_CACHE": "0"}, {"HK_NODE_CACHE": "0", "HK_DYNAMIC_SEGMENTS": "0"},
                {"HK_SMALL_PASS": "0"}, {"HK_SMALL_PASS": "0", "HK_WAVES_PER_CU": "2"}, {"HK_SMALL_PASS_WAVES": "4"}, {"HK_SMALL_PASS_WAVES": "16"},
                {"HK_SMALL_PASS_FUSED": "0"}, {"HK_SMALL_PASS_FUSED": "0", "HK_SMALL_PASS_WAVES": "8"},      # (the Cornell film's default is k_small_pass: the whole pass in one launch)
                {"HK_SHADOW_FINAL": "0"}, {"HK_SHADOW_FINAL": "0", "HK_SMALL_PASS_FUSED": "0", "HK_DYNAMIC_SEGMENTS": "1"},      # (round 6: the 60-byte shadow records — weights and path slot stored, the division in k_shadow — against the slim ones)
                {"HK_ESCAPED_UNROLL": "2"}, {"HK_ESCAPED_UNROLL": "2", "HK_SMALL_PASS_FUSED": "0"},      # (two escaped paths per lane and iteration)
                {"HK_LEAN_RECORDS": "0"}, {"HK_LEAN_RECORDS": "0", "HK_SMALL_PASS_FUSED": "0", "HK_DYNAMIC_SEGMENTS": "1"}, {"HK_LEAN_RECORDS": "0", "HK_SHADOW_FINAL": "0"},      # (round 6: 8-byte meta words and stored depth-0 origins against the lean records)
                {"HK_DYNAMIC_SEGMENTS": "1", "HK_TICKET_SHARE": "1"}, {"HK_DYNAMIC_SEGMENTS": "1", "HK_TICKET_SHARE": "0", "HK_WAVES_PER_CU": "3"}):     # (small passes: one segment per wave, no work lists / tickets — the default of a film this size)
        got = run(env)
        assert np.array_equal(ref.view(np.uint32), got.view(np.uint32)), env
    if which in ("slab", "cloud", "cloud_grid"):   # the shapes of the tracking loops (how many cheap steps per round, when idle lanes refill) are scheduling too
        for env in ({"HK_DELTA_ADVANCE": "1", "HK_TRACK_REFILL_IDLE": "1"}, {"HK_DELTA_ADVANCE": "9", "HK_TRACK_REFILL_IDLE": "40"},
                    {"HK_TRACK_ADVANCE": "1", "HK_SHADOW_TRACK_BATCH": "1", "HK_WALK_REFILL_IDLE": "1", "HK_SHADOW_FEED_ROUNDS": "1"},
                    {"HK_TRACK_ADVANCE": "7", "HK_SHADOW_TRACK_BATCH": "3", "HK_WALK_REFILL_IDLE": "33", "HK_SHADOW_FEED_ROUNDS": "5"}):
            got = run(env)
            assert np.array_equal(ref.view(np.uint32), got.view(np.uint32)), env
    if which in ("cloud", "cloud_grid"):
        # round 4: the tracking kernels with a per-wave ray pool in LDS (dense set-up / cast phases) against the per-lane refill kernels
        # they replace — k_track_pool / k_track_flat / k_track<GREY>, k_walk_pool (both pool sizes) / k_shadow_walk<GREY> — and under
        # odd segment counts (segments that end inside a phase, several segments per phase)
        for env in ({"HK_GREY_COMPACT": "0"}, {"HK_GREY_COMPACT": "0", "HK_WAVES_PER_CU": "5"},      # (the pool kernels on the full records: r_u stored, r_l four floats)
                    {"HK_TRACK_POOL": "0"}, {"HK_WALK_POOL": "0"}, {"HK_WALK_POOL": "2"}, {"HK_GREY_FLAT": "0"}, {"HK_TRACK_POOL": "0", "HK_WALK_POOL": "0", "HK_WAVES_PER_CU": "3"},
                    {"HK_WAVES_PER_CU": "3"}, {"HK_WAVES_PER_CU": "29", "HK_DELTA_ADVANCE": "1", "HK_TRACK_ADVANCE": "1"}, {"HK_WALK_POOL": "2", "HK_WAVES_PER_CU": "1"}, {"HK_WAVES_PER_CU": "0"}):   # (the segment count is sticky in the context: back to the default)
            got = run(env)
            assert np.array_equal(ref.view(np.uint32), got.view(np.uint32)), env
    if which == "cloud":
        # the shadow walk split into cast and tracking kernels with global queues in between: the same arithmetic per ray
        got = run({"HK_WALK_SPLIT": "1"})
        assert np.array_equal(ref.view(np.uint32), got.view(np.uint32)), "HK_WALK_SPLIT"
        got = run({"HK_WALK_SPLIT": "1", "HK_WAVES_PER_CU": "3", "HK_SHADOW_TRACK_BATCH": "2"})
        assert np.array_equal(ref.view(np.uint32), got.view(np.uint32)), "HK_WALK_SPLIT, 3 segments per CU"
        # the GREY kernels against the general ones: every decision is taken on the same arithmetic (identical paths), the weights
        # differ by the roundings of x * (T / T[1]) that the GREY code does not perform
        got = run({"HK_GREY": "0"})
        assert np.allclose(ref, got, rtol=2e-5, atol=1e-7), float(np.abs(ref - got).max())
    for k in names:
        knobs.delenv(k, raising=False)
    # leave the context's sticky knobs at their defaults for the tests that follow
    knobs.setenv("HK_OVERLAP", "1")
    knobs.setenv("HK_WAVES_PER_CU", "0")
    run({"HK_OVERLAP": "1", "HK_WAVES_PER_CU": "0"})


def test_node_cache_partial_tree(hk, gpu_ctx, oracle, knobs):
    """A tree LARGER than the LDS node cache but at most 16 deep: lanes at cached levels read LDS, lanes below read global memory, in
    the same wave (k_trace_lean 1536 nodes, k_shadow 112).  The film must equal, bit for bit, the one rendered with HK_NODE_CACHE=0,
    and agree with the oracle's frame."""
    import ctypes as C
    from hikari_jl_amd import geometry as G
    R = hk.RGBSpectrum
    w, h = 48, 40
    s = hk.Scene()
    s.push(G.sphere((0.0, 0.0, 3.0), 1.0, nvertices=64), hk.MatteMaterial(Kd=R(0.7, 0.5, 0.3)))
    s.push(G.quad((-3, -1.0, 0.5), (-3, -1.0, 6), (3, -1.0, 6), (3, -1.0, 0.5), normal=(0, 1, 0)), hk.MatteMaterial(Kd=R(0.5)))
    s.push(hk.PointLight((2.0, 3.0, 0.5), R(20.0)))
    s.sync()
    cam = hk.PerspectiveCamera((0, 0.3, 0), (0, 0, 3), hk.Film((w, h)), fov=50.0)
    n_nodes, depth = C.c_int32(), C.c_int32()
    hk._lib.lib().hk_scene_bvh_info(hk.scene_handle(gpu_ctx, s), C.byref(n_nodes), None, C.byref(depth))
    assert n_nodes.value > 1536 and depth.value <= 16, (n_nodes.value, depth.value)

    def run(cache):
        knobs.setenv("HK_NODE_CACHE", cache)
        film = hk.Film((w, h))
        vp = hk.VolPath(max_depth=5, samples=64)
        vp._ensure(film)
        vp.clear()
        vp.render_samples(s, film, cam, 16, first=1)
        acc, img = vp.read_accumulators(film).copy(), film.framebuffer.copy()
        vp.close()
        return acc, img

    acc1, img1 = run("1")
    acc0, _ = run("0")
    knobs.delenv("HK_NODE_CACHE", raising=False)
    assert np.array_equal(acc0.view(np.uint32), acc1.view(np.uint32))
    oacc, _ = oracle.OracleScene(s).render(hk.integrator_params(max_depth=5, samples=64), cam, w, h, 16)
    ref = oracle.finalize(oacc, w, h)
    rel_mse, frac = frame_metrics(img1, ref)
    assert img1.max() > 0.05 and rel_mse <= 1e-3 and frac >= 0.99, (rel_mse, frac)


def test_medium_furnace_gain_q30(hk):
    """The HIP path against the closed form of tests/test_independent_pins.py::test_medium_white_furnace_and_single_scatter: a
    non-absorbing isotropic slab of optical thickness 1.5 inside a constant environment returns 1.387 times the environment (the
    reference's double-counted direct light behind a specular medium boundary, quirk Q30) — the value of an independent float64
    random walk, not of the oracle."""
    from hikari_jl_amd import scenes
    from hikari_jl_amd import geometry as G
    R = hk.RGBSpectrum
    w = h = 32

    def frame(medium):
        _, film, cam = scenes.slab_scene(w, h, medium)
        s = hk.Scene()
        if medium is not None:
            s.push(G.rect3f((-2.5, -2.6, 1.0), (5.0, 5.2, 1.0)), hk.MediumInterface(hk.GlassMaterial(Kr=R(0.0), Kt=R(1.0), index=1.0), inside=medium, outside=None))
        s.push(hk.EnvironmentLight(hk.EnvironmentMap(np.full((16, 16, 3), 6e-5, np.float32)), R(1.0, 0.9, 0.8)))
        s.sync()
        vp = hk.VolPath(max_depth=64, samples=256)
        vp._ensure(film)
        vp.clear()
        vp.render_samples(s, film, cam, 256, first=1)
        img = film.framebuffer.copy()
        vp.close()
        return img[h // 4: 3 * h // 4, w // 4: 3 * w // 4].mean(axis=(0, 1))

    ref = frame(None)
    gain = frame(hk.HomogeneousMedium(sigma_a=R(0.0), sigma_s=R(1.5), g=0.0)) / ref
    assert np.allclose(gain, 1.387, rtol=0.012), gain
    # null collisions (same CPU test, walk_gain_tracked): the ratio-tracked shadow ray's T / (r_l + r_u) weight raises the factor with
    # the slack of the majorant — 1.475 when one global majorant cell is twice the density (a corner voxel doubles it), 1.449 for a
    # depth ramp through the NanoVDB tree under one global cell
    nv_bounds = ((-2.5, -2.6, 1.0), (2.5, 2.6, 2.0))
    spike = np.ones((16, 16, 16), np.float32)
    spike[0, 0, 0] = 2.0
    gain = frame(hk.NanoVDBMedium(spike, bounds=nv_bounds, sigma_a=R(0.0), sigma_s=R(1.5), g=0.0, majorant_res=(1, 1, 1))) / ref
    assert np.allclose(gain, 1.475, rtol=0.012), gain
    ramp = np.tile((0.25 + 1.5 * (np.arange(16) + 0.5) / 16)[None, None, :], (16, 16, 1)).astype(np.float32)
    gain = frame(hk.NanoVDBMedium(ramp, bounds=nv_bounds, sigma_a=R(0.0), sigma_s=R(1.5), g=0.0, majorant_res=(1, 1, 1))) / ref
    assert np.allclose(gain, 1.449, rtol=0.012), gain


def test_surface_furnace_q31(hk):
    """The HIP path on the closed emissive box of tests/test_independent_pins.py::test_surface_furnace_radiosity_closed_form: an
    independent float64 random walk of the reference's estimator gives 1.3896 Le for rho = 0.3 (2.7 % below Le / (1 - rho): the
    emission MIS evaluates the light-choice pmf at the hit point, quirk Q31) — the frame must land there, not on the oracle's word."""
    import sys, os
    sys.path.insert(0, os.path.dirname(__file__))
    from test_independent_pins import _emissive_box

    def mean(rho, spp):
        s, film, cam = _emissive_box(hk, rho)
        vp = hk.VolPath(max_depth=40, samples=4096, max_component_value=1e9)
        vp._ensure(film)
        vp.clear()
        vp.render_samples(s, film, cam, spp, first=1)
        img = film.framebuffer.copy()
        vp.close()
        return img.mean(axis=(0, 1))

    base = mean(0.0, 256)
    got = (mean(0.3, 512) / base).mean()
    assert abs(got - 1.3896) < 0.006, got


def test_nested_media_shadow_on_device(hk):
    """The HIP shadow walk through nested medium transitions against the closed form of
    tests/test_independent_pins.py::test_nested_media_shadow_transmittance (exp(-sigma_A (l_A - l_B) - sigma_B l_B))."""
    import sys, os
    sys.path.insert(0, os.path.dirname(__file__))
    from test_independent_pins import _nested_media_scene, _nested_media_expected

    def mean(sa, sb):
        s, cam = _nested_media_scene(hk, sa, sb)
        film = hk.Film((8, 8))
        vp = hk.VolPath(max_depth=4, samples=2048, max_component_value=1e9)
        vp._ensure(film)
        vp.clear()
        vp.render_samples(s, film, cam, 2048, first=1)
        img = film.framebuffer.copy()
        vp.close()
        return img.mean(axis=(0, 1))

    base = mean(None, None)
    for sa, sb in ((0.5, 2.0), (1.2, 0.3)):
        got = mean(sa, sb) / base
        assert np.allclose(got, _nested_media_expected(sa, sb), rtol=0.015), (sa, sb, got)


def test_specular_and_absorption_closed_forms_on_device(hk, oracle):
    """The HIP path against the closed forms of tests/test_independent_pins.py directly (not through the oracle's frames): an emitter in
    a mirror is Kr Le, through a glass slab (1 - R) / (1 + R) Le, behind a heterogeneous absorbing NanoVDB box exp(-integral sigma_a)."""
    from hikari_jl_amd import geometry as G
    R = hk.RGBSpectrum
    wh = 16

    def view(extra, spp, fov=6.0, max_depth=40):
        s = hk.Scene()
        em = G.quad((-6, -6, 6), (-6, 6, 6), (6, 6, 6), (6, -6, 6), normal=(0, 0, -1))
        s.push(em, hk.MediumInterface(hk.MatteMaterial(Kd=R(0.0)), emission=hk.Emissive(Le=R(0.2), scale=1.0, two_sided=True)))
        extra(s)
        s.sync()
        film = hk.Film((wh, wh))
        cam = hk.PerspectiveCamera((0, 0, 0), (0, 0, 1), film, fov=fov)
        vp = hk.VolPath(max_depth=max_depth, samples=4096, max_component_value=1e9)
        vp._ensure(film)
        vp.clear()
        vp.render_samples(s, film, cam, spp, first=1)
        m = film.framebuffer.mean(axis=(0, 1)).astype(np.float64)
        vp.close()
        return m, s

    base, _ = view(lambda s: None, 64)

    def mirror(s):
        s.push(G.quad((-2, -2, 3), (2, -2, 3), (2, 2, 3), (-2, 2, 3), normal=(0, 0, -1)), hk.MirrorMaterial(Kr=R(0.8)))
        em2 = G.quad((-9, -9, -6), (9, -9, -6), (9, 9, -6), (-9, 9, -6), normal=(0, 0, 1))
        s.push(em2, hk.MediumInterface(hk.MatteMaterial(Kd=R(0.0)), emission=hk.Emissive(Le=R(0.2), scale=1.0, two_sided=True)))

    got, _ = view(mirror, 64)
    assert np.allclose(got / base, 0.8, rtol=2e-3), got / base
    n = 1.5
    Rf = ((n - 1) / (n + 1)) ** 2
    got, _ = view(lambda s: s.push(G.rect3f((-3, -3, 2.0), (6, 6, 0.5)), hk.GlassMaterial(Kr=R(1.0), Kt=R(1.0), index=n)), 2048)
    assert np.allclose(got / base, (1 - Rf) / (1 + Rf), rtol=0.008), (got / base, (1 - Rf) / (1 + Rf))

    rng = np.random.default_rng(21)
    dens = (0.2 + 1.6 * rng.random((12, 10, 14))).astype(np.float32)
    dens[3:7, 2:6, 4:9] = 0.0
    bounds = ((-1.0, -1.0, 2.0), (1.0, 1.0, 4.0))
    med = hk.NanoVDBMedium(dens, bounds=bounds, sigma_a=R(1.0), sigma_s=R(0.0), majorant_res=(5, 4, 6))
    got, s = view(lambda s: s.push(G.rect3f(bounds[0], (2.0, 2.0, 2.0)), hk.MediumInterface(hk.GlassMaterial(Kr=R(0.0), Kt=R(1.0), index=1.0), inside=med, outside=None)), 2048, fov=14.0)
    osc = oracle.OracleScene(s)
    pq = hk.integrator_params(max_depth=24, samples=4096, max_component_value=1e9)
    camq = hk.PerspectiveCamera((0, 0, 0), (0, 0, 1), hk.Film((wh, wh)), fov=14.0)
    r2 = np.random.default_rng(5)
    cs = oracle.camera_samples(pq, camq, wh, wh, r2.integers(1, wh + 1, 1500).astype(np.int32), r2.integers(1, wh + 1, 1500).astype(np.int32),
                               r2.integers(1, 500, 1500).astype(np.int32)).astype(np.float64)
    fw, dirs = cs[:, 8], cs[:, 12:15]
    ts = np.linspace(0.0, 7.0, 1401)
    T = []
    for dd in dirs:
        P = (dd[None, :] * ts[:, None]).astype(np.float32)
        inside = np.all((P > np.array(bounds[0])) & (P < np.array(bounds[1])), axis=1)
        sig = osc.medium(0, 0, P, np.full((len(P), 4), 550.0, np.float32))[:, 0].astype(np.float64) * inside
        T.append(np.exp(-float(((sig[1:] + sig[:-1]) * 0.5 * np.diff(ts)).sum())))
    osc.close()
    want = float((fw * np.array(T)).sum() / fw.sum())
    assert np.allclose(got / base, want, rtol=0.02), (got / base, want)

"""SURVEY §8 row a24 — EnvironmentLight: equal-area map, Distribution2D importance sampling, escaped-ray MIS.
(textures/environment_map.jl:78-229, 290-371; sampler/sampling.jl:207-361; physical-wavefront/lights.jl:158-190, 336-347,
408-467; volpath/intersection.jl:622-678.)"""
import numpy as np
import pytest


def _unit(v):
    return (v / np.linalg.norm(v, axis=1, keepdims=True)).astype(np.float32)


def _env_scene(hk, data, rotation=None, scale=None, extra_ambient=False):
    from hikari_jl_amd import geometry as G
    s = hk.Scene()
    s.push(G.rect3f((-1, -1, -0.01), (2, 2, 0.01)), hk.MatteMaterial(Kd=hk.RGBSpectrum(0.6)))
    s.push(hk.EnvironmentLight(hk.EnvironmentMap(data, rotation), scale))
    if extra_ambient:
        s.push(hk.AmbientLight(hk.RGBSpectrum(0.2, 0.3, 0.4)))
    s.sync()
    return s


def _inputs(n, seed=3):
    rng = np.random.default_rng(seed)
    p = (rng.random((n, 3)) * 2 - 1).astype(np.float32)
    u = rng.random((n, 3), dtype=np.float32)
    d = _unit(rng.normal(size=(n, 3)))
    lam = (360 + 470 * rng.random((n, 4))).astype(np.float32)
    return p, u, d, lam


def test_distribution2d_tables(hk):
    """Distribution2D(func) (sampling.jl:207-262): cdf rows end at 1, marginal_func == row integrals, zero rows fall back to
    the uniform cdf, sequential Float32 accumulation."""
    rng = np.random.default_rng(1)
    f = rng.random((5, 7)).astype(np.float32)
    f[2] = 0
    D = hk.Distribution2D(f)
    assert D.nu == 7 and D.nv == 5
    assert np.array_equal(D.conditional_func, f)
    assert np.allclose(D.conditional_cdf[:, -1], 1.0) and (D.conditional_cdf[:, 0] == 0).all()
    assert np.array_equal(D.conditional_cdf[2], (np.arange(8, dtype=np.float32) / np.float32(7)))
    ref = np.zeros(8, np.float32)
    for u in range(1, 8):
        ref[u] = ref[u - 1] + f[0, u - 1] / np.float32(7)
    assert D.conditional_func_int[0] == ref[7] and np.array_equal(D.conditional_cdf[0], ref / ref[7])
    assert np.allclose(D.conditional_func_int, f.mean(axis=1), rtol=1e-6)
    assert np.isclose(D.marginal_func_int, f.mean(), rtol=1e-6) and D.marginal_cdf[-1] == 1.0


def test_rotation_matrix_is_julias_column_major_fill(hk):
    """rotation_matrix(90, z) (environment_map.jl:52-66): Mat3f(...) fills columns, so M[2,1] = -s*a3 + ... = -1."""
    M = hk.rotation_matrix(90.0, (0, 0, 1))
    assert np.allclose(M, [[0, 1, 0], [-1, 0, 0], [0, 0, 1]], atol=1e-6)
    assert np.allclose(M @ M.T, np.eye(3), atol=1e-6)


def test_equal_area_mapping_and_sampling_consistency(hk, oracle):
    """(1) sampled direction -> pdf_li gives back the sampling pdf; (2) the pdf integrates to 1 over the sphere;
    (3) a constant map samples uniformly with pdf 1/(4 pi) and radiance = uplift_illuminant(scale * c);
    (4) the sampled radiance is the nearest texel, the escaped radiance the bilinear one."""
    sky = hk.analytic_sky(32)
    rot = hk.rotation_matrix(35.0, (0.2, 1.0, 0.4))
    s = _env_scene(hk, sky, rot, hk.RGBSpectrum(0.5, 0.6, 0.7))
    osc = oracle.OracleScene(s)
    n = 40000
    p, u, d, lam = _inputs(n)
    S = osc.light(0, 1, p, u, lam)
    assert np.isfinite(S).all() and (S[:, 3] > 0).all() and (S[:, 11] == 0).all()
    assert np.allclose(np.linalg.norm(S[:, 0:3], axis=1), 1.0, atol=2e-5)
    assert np.allclose(S[:, 8:11], p + np.float32(1e6) * S[:, 0:3], rtol=1e-6, atol=1e-2)
    E = osc.light(1, 1, p, S[:, 0:3].copy(), lam)
    same_texel = np.isclose(E[:, 4], S[:, 3], rtol=1e-5)
    assert same_texel.mean() > 0.97            # the polynomial atan puts a few round trips into the neighbouring texel
    U = osc.light(1, 1, p, d, lam)
    assert abs(U[:, 4].mean() * 4 * np.pi - 1.0) < 0.05          # E_uniform[pdf] * 4 pi == integral of pdf == 1
    # importance sampling works: E_pdf[Le/pdf] == E_uniform[Le] * 4 pi (per wavelength), with far lower variance
    lhs = (S[:, 4:8] / S[:, 3:4]).mean(0)
    rhs = U[:, 0:4].mean(0) * 4 * np.pi
    assert np.allclose(lhs, rhs, rtol=0.08)
    osc.close()
    const = np.full((16, 16, 3), 0.25, np.float32)
    s2 = _env_scene(hk, const, None, hk.RGBSpectrum(2.0))
    osc2 = oracle.OracleScene(s2)
    S2 = osc2.light(0, 1, p, u, lam)
    assert np.allclose(S2[:, 3], 1 / (4 * np.pi), rtol=1e-5)
    want = oracle.uplift(2, np.tile([[0.5, 0.5, 0.5]], (n, 1)), lam)
    assert np.allclose(S2[:, 4:8], want, rtol=1e-5)
    E2 = osc2.light(1, 1, p, d, lam)
    assert np.allclose(E2[:, 0:4], want, rtol=1e-4) and np.allclose(E2[:, 4], 1 / (4 * np.pi), rtol=1e-5)
    zs = S2[:, 2]
    assert abs(zs.mean()) < 0.02 and abs((zs ** 2).mean() - 1 / 3) < 0.02     # uniform on the sphere
    osc2.close()


def test_sky_scene_on_the_oracle(hk, oracle):
    """Config-3-shaped scene end to end: finite, sun + sky both contribute, env MIS path exercised (escaped rays after a
    diffuse bounce), deterministic."""
    from hikari_jl_amd import scenes
    w = h = 40
    s, film, cam = scenes.sky_scene(w, h, env_res=32, tess=16)
    p = hk.integrator_params(max_depth=6, samples=4)
    osc = oracle.OracleScene(s)
    acc, st = osc.render(p, cam, w, h, 4)
    img = oracle.finalize(acc, w, h)
    assert np.isfinite(img).all() and (img >= 0).all() and img.mean() > 0.05
    acc2, _ = osc.render(p, cam, w, h, 4)
    assert np.array_equal(acc, acc2)
    s_nosun, _, _ = scenes.sky_scene(w, h, env_res=32, tess=16, sun=False)
    acc3, _ = oracle.OracleScene(s_nosun).render(p, cam, w, h, 4)
    assert oracle.finalize(acc3, w, h).mean() < img.mean()


def test_hosek_wilkie_bake(hk):
    """sunsky_to_envlight (lights/sun_sky.jl:358-434): 13 wavelengths 320..720 nm -> XYZ / CIE_Y_integral -> linear sRGB >= 0,
    equal-area map, EnvironmentLight scale = intensity / 10567, SunLight = RGB(5, 4.75, 4.25) * intensity along -dir with the
    RGB-constructor scale 1/10567.  Physical sanity of the baked sky: bluer at the zenith than at the horizon, brighter on the
    sun's side at equal elevation, whiter with higher turbidity, zero below the horizon when the ground is off."""
    env, sun = hk.sunsky_to_envlight((1, 2, 9), intensity=2.0, turbidity=3.0, ground_enabled=False, resolution=64)
    d = env.env_map.data[..., :3]
    assert d.shape == (64, 64, 3) and np.isfinite(d).all() and (d >= 0).all() and d.max() < 5.0
    assert np.isclose(env.scale_rgb.c[0], 2.0 / 10567.0, rtol=1e-6) and np.isclose(sun.scale, 1.0 / 10567.0, rtol=1e-6)
    n = np.array([1, 2, 9]) / np.linalg.norm([1, 2, 9])
    assert np.allclose(sun.direction, -n, atol=1e-6)
    # SunLight(RGB(5, 4.75, 4.25) * intensity, ...) bakes rgb_illuminant_spectrum: scale = 2 max(rgb), poly of rgb / scale
    from hikari_jl_amd.tables import rgb_to_spectrum
    assert np.isclose(sun.i.scale, 20.0, rtol=1e-6)
    assert np.allclose(sun.i.poly, rgb_to_spectrum(np.float32(0.5), np.float32(0.475), np.float32(0.425)), rtol=1e-6)
    from hikari_jl_amd.envmap import equal_area_square_to_sphere
    c = (np.arange(64) + 0.5) / 64
    uu, vv = np.meshgrid(c, c)
    x, y, z = equal_area_square_to_sphere(uu, vv)
    zen, hor = d[(z > 0.9)], d[(z > 0.02) & (z < 0.15)]
    assert (zen[:, 2] / zen[:, 0]).mean() > (hor[:, 2] / hor[:, 0]).mean()            # blue ratio
    cs = x * n[0] + y * n[1] + z * n[2]
    band = (z > 0.3) & (z < 0.6)                                                       # same elevation: brighter on the sun's side
    assert d[band & (cs > np.median(cs[band]))][:, 1].mean() > d[band & (cs < np.median(cs[band]))][:, 1].mean()
    hazy, _ = hk.sunsky_to_envlight((1, 2, 9), intensity=2.0, turbidity=8.0, ground_enabled=False, resolution=64)
    hz = hazy.env_map.data[..., :3][z > 0.9]
    assert (hz[:, 2] / hz[:, 0]).mean() < (zen[:, 2] / zen[:, 0]).mean()                # haze whitens the zenith
    g_env, _ = hk.sunsky_to_envlight((1, 2, 9), ground_albedo=hk.RGBSpectrum(0.5, 0.4, 0.3), ground_enabled=True, resolution=32)
    gd = g_env.env_map.data
    c32 = (np.arange(32) + 0.5) / 32
    _, _, z32 = equal_area_square_to_sphere(*np.meshgrid(c32, c32))
    assert np.allclose(gd[z32 < -0.05][:, :3], np.array([0.5, 0.4, 0.3], np.float32) * np.float32(0.3))


def test_metal_presets(hk):
    """Gold()/Silver()/Copper()/Aluminum()/Brass() (uber-material.jl:455-520) carry the measured eta/k spectra."""
    g = hk.Gold(roughness=0.01)
    assert g.kind == hk._abi.HK_MAT_CONDUCTOR and g.eta.lambdas.size == 56 and g.k.values.size == 56
    assert np.isclose(g.eta.lambdas[0], 298.75705) and np.isclose(g.eta.lambdas[-1], 885.60126, rtol=1e-6)
    assert g.eta.values[(g.eta.lambdas > 600) & (g.eta.lambdas < 700)].mean() < 0.3          # Au: n << 1 in the red
    assert hk.Brass().eta.lambdas.size == 61
    d = hk.ConductorMaterial()
    assert d.eta.c[:3] == (np.float32(0.2),) * 3 and d.roughness == 0.1


# --------------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
def test_environment_light_pointwise_parity(hk, oracle, gpu_ctx):
    sky = hk.analytic_sky(64)
    s = _env_scene(hk, sky, hk.rotation_matrix(35.0, (0.2, 1.0, 0.4)), hk.RGBSpectrum(0.5, 0.6, 0.7), extra_ambient=True)
    osc = oracle.OracleScene(s)
    n = 100000
    p, u, d, lam = _inputs(n, seed=17)
    u[:200, 0] = 0.0
    u[200:400, 1] = 0.0
    u[400:600] = np.float32(1.0 - 2 ** -24)
    d[:6] = np.array([[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1]], np.float32)   # poles / seams
    d[6:8] = np.array([[0, 0, 1 + 2.0 ** -23], [0, 0, -(1 + 2.0 ** -23)]], np.float32)                      # quirk Q35: |z| one ulp above 1
    sh = hk.scene_handle(gpu_ctx, s)
    L = hk._lib.lib()
    PF = hk._abi.PF
    for mode, x in ((0, u), (1, d)):
        for light in ((1, 2) if mode == 0 else (1,)):
            ref = osc.light(mode, light, p, x, lam)
            out = np.zeros((n, 12), np.float32)
            hk._lib.check(L.hk_test_light(gpu_ctx.h, sh, mode, light, n, *[a.ctypes.data_as(PF) for a in (p, x, lam, out)]), "hk_test_light")
            assert np.isfinite(out).all()
            close = np.isclose(out, ref, rtol=3e-5, atol=1e-6).all(axis=1)
            assert close.mean() >= 0.9995, (mode, light, close.mean())
    osc.close()


@pytest.mark.gpu
@pytest.mark.parametrize("sun,analytic", [(True, False), (False, False), (True, True)])
def test_sky_scene_frame_parity(hk, oracle, sun, analytic):
    """Config 3 stand-in (glass sphere + rough-gold slab + EnvironmentLight [+ SunLight]), strict frame parity."""
    from hikari_jl_amd import scenes
    from test_gpu_parity import frame_metrics
    w = h = 64
    s, film, cam = scenes.sky_scene(w, h, env_res=64, tess=24, sun=sun, analytic=analytic)
    kw = dict(max_depth=8, samples=8)
    p = hk.integrator_params(**kw)
    acc, ost = oracle.OracleScene(s).render(p, cam, w, h, kw["samples"])
    ref = oracle.finalize(acc, w, h)
    vp = hk.VolPath(**kw)
    vp(s, film, cam)
    rel_mse, frac_ok = frame_metrics(film.framebuffer, ref)
    assert np.isfinite(film.framebuffer).all()
    assert rel_mse <= 1e-3 and frac_ok >= 0.99, (rel_mse, frac_ok)
    st = vp.stats()
    assert abs(int(st.rays_closest) - int(ost.rays_closest)) <= 0.002 * ost.rays_closest + 8
    assert abs(int(st.rays_shadow) - int(ost.rays_shadow)) <= 0.002 * ost.rays_shadow + 8
    vp.close()


@pytest.mark.gpu
def test_sky_scene_bench_configuration(hk, oracle):
    """Config 3 exactly as bench.py --config sky builds it — 512^2 equal-area environment map, default tessellation, depth 12 — first
    against the oracle on a small film (same seed, strict), then at the bench's 800 x 800 through size-independent properties: finite,
    the rays per path of the full frame equal those of the small one within 1 %, and two sample-index shards add up to the unsharded film."""
    from hikari_jl_amd import scenes
    from test_gpu_parity import frame_metrics
    kw = dict(max_depth=12, samples=256)
    w = h = 80
    s, film, cam = scenes.sky_scene(w, h, env_res=512)
    p = hk.integrator_params(**kw)
    osc = oracle.OracleScene(s)
    acc, ost = osc.render(p, cam, w, h, 4)
    osc.close()
    ref = oracle.finalize(acc, w, h)
    vp = hk.VolPath(**kw)
    vp._ensure(film)
    vp.clear()
    vp.reset_stats()
    vp.render_samples(s, film, cam, 4, first=1)
    st = vp.stats()
    rel_mse, frac_ok = frame_metrics(film.framebuffer, ref)
    assert rel_mse <= 1e-3 and frac_ok >= 0.99, (rel_mse, frac_ok)
    assert abs(int(st.rays_closest) - int(ost.rays_closest)) <= 0.002 * ost.rays_closest + 8
    small_rays_per_path = (int(st.rays_closest) + int(st.rays_shadow)) / (w * h * 4)
    vp.close()

    W = H = 800
    s2, film2, cam2 = scenes.sky_scene(W, H, env_res=512)
    vp = hk.VolPath(**kw)
    vp._ensure(film2)
    vp.clear()
    vp.reset_stats()
    vp.render_samples(s2, film2, cam2, 16, first=1)
    st = vp.stats()
    full = vp.read_accumulators(film2).copy()
    assert np.isfinite(full).all() and full.max() > 0
    rays_per_path = (int(st.rays_closest) + int(st.rays_shadow)) / (W * H * 16)
    assert abs(rays_per_path / small_rays_per_path - 1.0) < 0.01, (rays_per_path, small_rays_per_path)
    vp.clear()
    vp.render_samples(s2, film2, cam2, 8, stride=2, first=1)
    vp.render_samples(s2, film2, cam2, 8, stride=2, first=2)
    both = vp.read_accumulators(film2)
    assert np.allclose(both, full, rtol=2e-4, atol=1e-5)      # the same 16 samples per pixel, summed in another order
    vp.close()

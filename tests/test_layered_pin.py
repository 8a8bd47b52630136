"""CoatedDiffuseMaterial and CoatedDiffuseTransmissionMaterial from a second source ("a wrong r_l after ... a CoatedDiffuse bounce"): tests/ref_layered_np.py
restates pbrt-v4's LayeredBxDF as the reference ports it (materials/spectral-eval.jl:815-1940) in scalar float32 NumPy and is compared
POINT BY POINT with the oracle's sample_bsdf_spectral / evaluate_bsdf_spectral on identical (wo, wi, n, lambda, u, uc).  The random walks
draw from a PCG32 seeded by hashes of the float bits of their inputs, so identical inputs give the same walk on both sides and the
comparison is exact up to the roundings of float32 transcendentals — except where such a rounding flips a branch of the walk (a
reflection / transmission choice, a roulette): those samples differ wholesale and are counted."""
import numpy as np
import pytest

import ref_layered_np as LN
import ref_volpath_np as R


def _unit(v):
    return (v / np.linalg.norm(v, axis=-1, keepdims=True)).astype(np.float32)


CASES = {
    # name: (kwargs of CoatedDiffuseMaterial, regularize)
    "smooth_no_medium": (dict(reflectance=(0.7, 0.4, 0.2), u_roughness=0.0, v_roughness=0.0, thickness=0.02, eta=1.5), False),
    "rough": (dict(reflectance=(0.5, 0.6, 0.3), u_roughness=0.25, v_roughness=0.25, thickness=0.05, eta=1.4), False),
    "rough_aniso_medium": (dict(reflectance=(0.6, 0.5, 0.5), u_roughness=0.3, v_roughness=0.1, thickness=0.3, eta=1.5, albedo=(0.7, 0.8, 0.6), g=0.4, max_depth=12, n_samples=2), False),
    "smooth_medium_regularized": (dict(reflectance=(0.3, 0.5, 0.7), u_roughness=0.0005, v_roughness=0.0005, thickness=0.2, eta=1.6, albedo=(0.5, 0.5, 0.9), g=-0.3, remap_roughness=False), True),
}


CDT_CASES = {
    "cdt_smooth": (dict(reflectance=(0.5, 0.3, 0.2), transmittance=(0.3, 0.5, 0.6), u_roughness=0.0, v_roughness=0.0, thickness=0.05, eta=1.5), False),
    "cdt_rough_medium": (dict(reflectance=(0.6, 0.6, 0.4), transmittance=(0.2, 0.3, 0.7), u_roughness=0.2, v_roughness=0.35, thickness=0.25, eta=1.45, albedo=(0.8, 0.6, 0.7), g=0.3,
                              max_depth=12, n_samples=2), False),
    "cdt_regularized": (dict(reflectance=(1.4, 0.2, 0.1), transmittance=(0.05, 0.9, 0.3), u_roughness=0.01, v_roughness=0.01, thickness=0.1, eta=1.6), True),
}
CASES.update(CDT_CASES)


def _params(hk, kw, lam, tables):
    if "transmittance" in kw:      # CoatedDiffuseTransmission: reflectance / transmittance clamped to [0, 1], the base's lobe chosen by their largest components
        cl = lambda c: np.clip(np.array(c, np.float32), 0, 1)
        rr, tt = cl(kw["reflectance"]), cl(kw["transmittance"])
        up = lambda c: R.eval_poly(R.F(tables.rgb_to_poly([float(x) for x in c]))[None], lam[None])[0]
        alb_rgb = np.array(kw.get("albedo", (0.0, 0.0, 0.0)), np.float32)
        remap = kw.get("remap_roughness", True)
        al = lambda r: np.float32(np.sqrt(np.float32(r))) if remap else np.float32(r)
        refl, trans = up(rr), up(tt)
        return LN.Coated(refl, up(alb_rgb), bool((alb_rgb != 0).any()), al(kw.get("u_roughness", 0.0)), al(kw.get("v_roughness", 0.0)), kw.get("eta", 1.5), kw.get("thickness", 0.01),
                         kw.get("g", 0.0), kw.get("max_depth", 10), kw.get("n_samples", 1), bottom=LN.DiffuseTransmissionBottom(refl, trans, rr.max(), tt.max()))
    rgb = lambda c: np.array(c if np.ndim(c) else (c, c, c), np.float32)
    refl = R.eval_poly(R.F(tables.rgb_to_poly(list(rgb(kw["reflectance"]))))[None], lam[None])[0]
    alb_rgb = rgb(kw.get("albedo", 0.0))
    albedo = R.eval_poly(R.F(tables.rgb_to_poly(list(alb_rgb)))[None], lam[None])[0]
    remap = kw.get("remap_roughness", True)
    al = lambda r: np.float32(np.sqrt(np.float32(r))) if remap else np.float32(r)
    return LN.Coated(refl, albedo, bool((alb_rgb != 0).any()), al(kw.get("u_roughness", 0.0)), al(kw.get("v_roughness", 0.0)), kw.get("eta", 1.5), kw.get("thickness", 0.01),
                     kw.get("g", 0.0), kw.get("max_depth", 10), kw.get("n_samples", 1))


@pytest.mark.parametrize("name", list(CASES))
def test_coated_diffuse_point_wise_against_the_numpy_restatement(hk, oracle, name):
    from hikari_jl_amd import geometry as G
    kw, regularize = CASES[name]
    Rg = hk.RGBSpectrum
    mk = dict(kw)
    for k in ("reflectance", "albedo", "transmittance"):
        if k in mk:
            mk[k] = Rg(*mk[k])
    s = hk.Scene()
    s.push(G.quad((-1, 0, -1), (1, 0, -1), (1, 0, 1), (-1, 0, 1), normal=(0, 1, 0)), (hk.CoatedDiffuseTransmissionMaterial if "transmittance" in mk else hk.CoatedDiffuseMaterial)(**mk))
    s.push(hk.PointLight((0, 2, 0), Rg(1.0)))
    s.sync()
    tables = R.Tables(hk.tables.load())
    rng = np.random.default_rng(sum(name.encode()))
    n = 160
    ns = _unit(rng.normal(size=(n, 3)))
    wo = _unit(rng.normal(size=(n, 3)))
    wi = _unit(rng.normal(size=(n, 3)))
    lam = (380 + 420 * rng.random((n, 4))).astype(np.float32)
    u = rng.random((n, 2)).astype(np.float32)
    uc = rng.random(n).astype(np.float32)
    osc = oracle.OracleScene(s)
    smp = osc.bsdf(0, 0, wo, wi, ns, lam, u, uc, regularize=regularize)
    evl = osc.bsdf(1, 0, wo, wi, ns, lam, u, uc)
    osc.close()

    def close(a, b, rt):
        a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
        return bool(np.all(np.abs(a - b) <= rt * np.maximum(np.abs(a), np.abs(b)) + 1e-12))

    bad_s = bad_e = n_valid = n_spec = 0
    for i in range(n):
        P = _params(hk, kw, lam[i], tables)
        got = LN.coated_sample(P, wo[i], ns[i], (u[i, 0], u[i, 1]), uc[i], regularize)
        r = smp[i]
        valid_ref = r[7] > 0 and np.any(r[3:7] != 0)
        if got is None:
            ok = not valid_ref
        else:
            w2, f2, p2, sp2, eta2 = got
            n_valid += 1
            n_spec += bool(sp2)
            ok = bool(valid_ref) and bool(np.abs(w2 - r[0:3]).max() <= 5e-6) and close(f2, r[3:7], 2e-4) and close([p2], r[7:8], 2e-4) and bool(sp2) == bool(r[8]) and close([eta2], r[9:10], 1e-6)
        bad_s += 0 if ok else 1
        f3, p3 = LN.coated_eval(P, wo[i], wi[i], ns[i])
        e = evl[i]
        bad_e += 0 if (close(f3, e[0:4], 5e-4) and close([p3], e[4:5], 5e-4)) else 1
    print("%s: %d samples (%d valid, %d specular paths), %d differ; %d evaluations, %d differ" % (name, n, n_valid, n_spec, bad_s, n, bad_e))
    assert n_valid >= n // 3
    assert bad_s <= max(2, n // 50) and bad_e <= max(2, n // 50)


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(CASES))
def test_device_coated_diffuse_point_wise_against_the_numpy_restatement(hk, gpu_ctx, name):
    """The HIP BSDF (hk_test_bsdf through the C-ABI) against the NumPy restatement directly — no oracle in between.  The device's log / exp /
    sin / cos are the hardware's (1 - 2 ulp): a few more walks take another branch than on the CPU."""
    from hikari_jl_amd import geometry as G
    kw, regularize = CASES[name]
    Rg = hk.RGBSpectrum
    mk = dict(kw)
    for k in ("reflectance", "albedo", "transmittance"):
        if k in mk:
            mk[k] = Rg(*mk[k])
    s = hk.Scene()
    s.push(G.quad((-1, 0, -1), (1, 0, -1), (1, 0, 1), (-1, 0, 1), normal=(0, 1, 0)), (hk.CoatedDiffuseTransmissionMaterial if "transmittance" in mk else hk.CoatedDiffuseMaterial)(**mk))
    s.push(hk.PointLight((0, 2, 0), Rg(1.0)))
    s.sync()
    tables = R.Tables(hk.tables.load())
    rng = np.random.default_rng(sum(name.encode()) + 7)
    n = 160
    ns = _unit(rng.normal(size=(n, 3)))
    wo = _unit(rng.normal(size=(n, 3)))
    wi = _unit(rng.normal(size=(n, 3)))
    lam = (380 + 420 * rng.random((n, 4))).astype(np.float32)
    u = rng.random((n, 2)).astype(np.float32)
    uc = rng.random(n).astype(np.float32)
    sh = hk.scene_handle(gpu_ctx, s)
    L = hk._lib.lib()
    PF = hk._abi.PF
    outs = []
    for mode, reg in ((0, regularize), (1, False)):
        out = np.zeros((n, 10), np.float32)
        hk._lib.check(L.hk_test_bsdf(gpu_ctx.h, sh, mode, 0, 1 if reg else 0, n, *[a.ctypes.data_as(PF) for a in (wo, wi, ns, lam, u, uc, out)]), "hk_test_bsdf")
        outs.append(out)
    smp, evl = outs

    def close(a, b, rt):
        a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
        return bool(np.all(np.abs(a - b) <= rt * np.maximum(np.abs(a), np.abs(b)) + 1e-12))

    bad_s = bad_e = 0
    for i in range(n):
        P = _params(hk, kw, lam[i], tables)
        got = LN.coated_sample(P, wo[i], ns[i], (u[i, 0], u[i, 1]), uc[i], regularize)
        r = smp[i]
        valid_dev = r[7] > 0 and np.any(r[3:7] != 0)
        if got is None:
            ok = not valid_dev
        else:
            w2, f2, p2, sp2, eta2 = got
            ok = bool(valid_dev) and bool(np.abs(w2 - r[0:3]).max() <= 1e-5) and close(f2, r[3:7], 1e-3) and close([p2], r[7:8], 1e-3) and bool(sp2) == bool(r[8]) and close([eta2], r[9:10], 1e-6)
        bad_s += 0 if ok else 1
        f3, p3 = LN.coated_eval(P, wo[i], wi[i], ns[i])
        bad_e += 0 if (close(f3, evl[i][0:4], 2e-3) and close([p3], evl[i][4:5], 2e-3)) else 1
    print("device vs restatement (coated diffuse, %s): %d samples, %d differ; %d evaluations, %d differ" % (name, n, bad_s, n, bad_e))
    assert bad_s <= n // 16 and bad_e <= n // 16


CC_CASES = {
    # (CoatedConductorMaterial kwargs, regularize): the four coating x base combinations, eta / k and reflectance modes, a medium in between
    "cc_smooth_smooth": (dict(interface_eta=1.5, conductor_u_roughness=0.0, conductor_v_roughness=0.0), False),
    "cc_smooth_rough": (dict(interface_eta=1.4, conductor_u_roughness=0.2, conductor_v_roughness=0.1, albedo=(0.8, 0.7, 0.9), thickness=0.2), False),
    "cc_rough_smooth": (dict(interface_u_roughness=0.15, interface_v_roughness=0.15, interface_eta=1.5, reflectance=(0.9, 0.6, 0.3)), False),
    "cc_rough_rough": (dict(interface_u_roughness=0.1, interface_v_roughness=0.3, interface_eta=1.6, conductor_u_roughness=0.25, conductor_v_roughness=0.25, albedo=(0.5, 0.9, 0.6), thickness=0.1), False),
    "cc_regularized": (dict(interface_u_roughness=0.0004, interface_v_roughness=0.0004, interface_eta=1.5, conductor_u_roughness=0.002, conductor_v_roughness=0.002, remap_roughness=False,
                            reflectance=(0.7, 0.7, 0.2)), True),
}


def _cc_params(kw, lam, tables):
    remap = kw.get("remap_roughness", True)
    al = lambda r: np.float32(np.sqrt(np.float32(r))) if remap else np.float32(r)
    up = lambda c: R.eval_poly(R.F(tables.rgb_to_poly([float(x) for x in c]))[None], lam[None])[0]
    if "reflectance" in kw:          # eta = 1, k = 2 sqrt(r) / sqrt(max(1 - r, 0) + 1e-6), r clamped to [0, 0.9999] (spectral-eval.jl:2921-2931)
        r = up(np.clip(np.array(kw["reflectance"], np.float32), 0, np.float32(0.9999)))
        ce = np.ones(4, np.float32)
        ck = (np.float32(2) * np.sqrt(r) / np.sqrt(np.maximum(np.float32(1) - r, np.float32(0)) + np.float32(1e-6))).astype(np.float32)
    else:                            # the host class's defaults (copper-like RGB eta / k), uplifted unbounded (eval_ior_spectral)
        ce = R.unbounded_eval(R.unbounded_poly(tables, [np.float32(v) for v in (0.2, 0.92, 1.1)]), R.F(lam)[None])[0]
        ck = R.unbounded_eval(R.unbounded_poly(tables, [np.float32(v) for v in (3.9, 2.45, 2.14)]), R.F(lam)[None])[0]
    alb = np.array(kw.get("albedo", (0.0, 0.0, 0.0)), np.float32)
    return LN.CoatedCond(kw.get("interface_eta", 1.5), al(kw.get("interface_u_roughness", 0.0)), al(kw.get("interface_v_roughness", 0.0)), al(kw.get("conductor_u_roughness", 0.0)),
                         al(kw.get("conductor_v_roughness", 0.0)), ce, ck, kw.get("thickness", 0.01), up(alb), bool((alb != 0).any()))


def _cc_compare(kw, regularize, smp, evl, wo, wi, ns, lam, u, uc, tables, rt_s, rt_e):
    def close(a, b, rt):
        a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
        return bool(np.all(np.abs(a - b) <= rt * np.maximum(np.abs(a), np.abs(b)) + 1e-12))
    n = len(uc)
    bad_s = bad_e = n_valid = n_spec = 0
    for i in range(n):
        P = _cc_params(kw, lam[i], tables)
        got = LN.cc_sample(P, wo[i], ns[i], (u[i, 0], u[i, 1]), uc[i], regularize)
        r = smp[i]
        valid_ref = r[7] > 0 and np.any(r[3:7] != 0)
        if got is None:
            ok = not valid_ref
        else:
            w2, f2, p2, sp2 = got
            ok_f = bool(p2 > 0) and bool(np.any(f2 != 0))          # (sample_spectral_material's caller drops pdf == 0 / black samples: both sides may report them differently)
            n_valid += ok_f
            n_spec += bool(sp2)
            ok = (not ok_f and not valid_ref) or (bool(valid_ref) and bool(np.abs(w2 - r[0:3]).max() <= 1e-5) and close(f2, r[3:7], rt_s) and close([p2], r[7:8], rt_s) and bool(sp2) == bool(r[8]))
        bad_s += 0 if ok else 1
        f3, p3 = LN.cc_eval(P, wo[i], wi[i], ns[i])
        bad_e += 0 if (close(f3, evl[i][0:4], rt_e) and close([p3], evl[i][4:5], rt_e)) else 1
    return n_valid, n_spec, bad_s, bad_e


def _cc_scene(hk, kw):
    from hikari_jl_amd import geometry as G
    Rg = hk.RGBSpectrum
    mk = dict(kw)
    for k in ("reflectance", "albedo"):
        if k in mk:
            mk[k] = Rg(*mk[k])
    s = hk.Scene()
    s.push(G.quad((-1, 0, -1), (1, 0, -1), (1, 0, 1), (-1, 0, 1), normal=(0, 1, 0)), hk.CoatedConductorMaterial(**mk))
    s.push(hk.PointLight((0, 2, 0), Rg(1.0)))
    s.sync()
    return s


def _cc_inputs(name, n, extra=0):
    rng = np.random.default_rng(sum(name.encode()) + extra)
    ns = _unit(rng.normal(size=(n, 3)))
    wo = _unit(rng.normal(size=(n, 3)))
    wi = _unit(rng.normal(size=(n, 3)))
    wi[::2] = _unit(wi[::2] + 1.5 * ns[::2] * np.sign((wo[::2] * ns[::2]).sum(1, keepdims=True)))      # half of the pairs on the same side (the lobe evaluates only there)
    lam = (380 + 420 * rng.random((n, 4))).astype(np.float32)
    return ns, wo, wi, lam, rng.random((n, 2)).astype(np.float32), rng.random(n).astype(np.float32)


@pytest.mark.parametrize("name", list(CC_CASES))
def test_coated_conductor_point_wise_against_the_numpy_restatement(hk, oracle, name):
    """CoatedConductorMaterial (spectral-eval.jl:2877-3412) is NOT a random walk in the reference: an analytic composition of the coating
    and the base, case by case — restated in ref_layered_np.cc_sample / cc_eval and compared with the oracle point by point."""
    kw, regularize = CC_CASES[name]
    s = _cc_scene(hk, kw)
    tables = R.Tables(hk.tables.load())
    n = 200
    ns, wo, wi, lam, u, uc = _cc_inputs(name, n)
    osc = oracle.OracleScene(s)
    smp = osc.bsdf(0, 0, wo, wi, ns, lam, u, uc, regularize=regularize)
    evl = osc.bsdf(1, 0, wo, wi, ns, lam, u, uc)
    osc.close()
    n_valid, n_spec, bad_s, bad_e = _cc_compare(kw, regularize, smp, evl, wo, wi, ns, lam, u, uc, tables, 2e-4, 5e-4)
    print("%s: %d samples (%d valid, %d specular), %d differ; %d evaluations (%d non-zero), %d differ" % (name, n, n_valid, n_spec, bad_s, n, int((evl[:, 4] > 0).sum()), bad_e))
    assert n_valid >= n // 3 and (name == "cc_smooth_smooth" or (evl[:, 4] > 0).sum() >= n // 4)
    assert bad_s <= 2 and bad_e <= 2


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(CC_CASES))
def test_device_coated_conductor_point_wise_against_the_numpy_restatement(hk, gpu_ctx, name):
    """the HIP BSDF against the restatement directly (hk_test_bsdf; no oracle in between)"""
    kw, regularize = CC_CASES[name]
    s = _cc_scene(hk, kw)
    tables = R.Tables(hk.tables.load())
    n = 200
    ns, wo, wi, lam, u, uc = _cc_inputs(name, n, 11)
    sh = hk.scene_handle(gpu_ctx, s)
    L = hk._lib.lib()
    PF = hk._abi.PF
    outs = []
    for mode, reg in ((0, regularize), (1, False)):
        out = np.zeros((n, 10), np.float32)
        hk._lib.check(L.hk_test_bsdf(gpu_ctx.h, sh, mode, 0, 1 if reg else 0, n, *[a.ctypes.data_as(PF) for a in (wo, wi, ns, lam, u, uc, out)]), "hk_test_bsdf")
        outs.append(out)
    n_valid, n_spec, bad_s, bad_e = _cc_compare(kw, regularize, outs[0], outs[1], wo, wi, ns, lam, u, uc, tables, 1e-3, 2e-3)
    print("device vs restatement (coated conductor, %s): %d samples, %d differ; %d evaluations, %d differ" % (name, n, bad_s, n, bad_e))
    assert bad_s <= n // 25 and bad_e <= n // 25

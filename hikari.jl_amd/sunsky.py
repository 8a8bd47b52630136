"""Hosek-Wilkie sun + sky as (EnvironmentLight, SunLight): host-side scene construction.

Follows src/lights/sun_sky.jl: hosek_cook_config :19-85, hosek_cook_radiance :87-125, hosek_radiance :127-140, HosekState
:146-163, hosek_spectral_radiance :165-190, _spectrum_to_xyz :319-339, sunsky_to_envlight :358-434, in Float64 like the
reference, vectorised over the texels of the equal-area map.  Coefficients: data/hosek_wilkie_sky.bin (see data/README.md)."""
import os

import numpy as np

from .envmap import EnvironmentLight, EnvironmentMap
from .lights import D65_PHOTOMETRIC, SunLight
from .materials import RGBSpectrum
from . import tables as _tables

f32 = np.float32
_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "hosek_wilkie_sky.bin")
_cache = {}


def _datasets():
    if "d" not in _cache:
        a = np.fromfile(_DATA, dtype=np.float64)
        assert a.size == 11 * 1080 + 11 * 120
        _cache["d"] = (a[:11 * 1080].reshape(11, 1080), a[11 * 1080:].reshape(11, 120))
    return _cache["d"]


def _bernstein5(t, c):
    s = 1.0 - t
    return s ** 5 * c[0] + 5.0 * s ** 4 * t * c[1] + 10.0 * s ** 3 * t ** 2 * c[2] + 10.0 * s ** 2 * t ** 3 * c[3] + 5.0 * s * t ** 4 * c[4] + t ** 5 * c[5]


def hosek_cook_config(dataset, turbidity, albedo, solar_elevation):
    it = int(np.clip(np.floor(turbidity), 1, 10))
    rem = turbidity - float(it)
    t = (solar_elevation / (np.pi / 2.0)) ** (1.0 / 3.0)
    cfg = np.zeros(9)

    def term(offset):   # 0-based offset of the 9x6 block
        blk = dataset[offset:offset + 54].reshape(6, 9)
        return np.array([_bernstein5(t, blk[:, i]) for i in range(9)])

    cfg += (1.0 - albedo) * (1.0 - rem) * term(9 * 6 * (it - 1))
    cfg += albedo * (1.0 - rem) * term(9 * 6 * 10 + 9 * 6 * (it - 1))
    if it < 10:
        cfg += (1.0 - albedo) * rem * term(9 * 6 * it)
        cfg += albedo * rem * term(9 * 6 * 10 + 9 * 6 * it)
    return cfg


def hosek_cook_radiance(dataset, turbidity, albedo, solar_elevation):
    it = int(np.clip(np.floor(turbidity), 1, 10))
    rem = turbidity - float(it)
    t = (solar_elevation / (np.pi / 2.0)) ** (1.0 / 3.0)
    res = (1.0 - albedo) * (1.0 - rem) * _bernstein5(t, dataset[6 * (it - 1):6 * (it - 1) + 6])
    res += albedo * (1.0 - rem) * _bernstein5(t, dataset[60 + 6 * (it - 1):60 + 6 * (it - 1) + 6])
    if it < 10:
        res += (1.0 - albedo) * rem * _bernstein5(t, dataset[6 * it:6 * it + 6])
        res += albedo * rem * _bernstein5(t, dataset[60 + 6 * it:60 + 6 * it + 6])
    return res


def hosek_radiance(c, theta, gamma):
    cg = np.cos(gamma)
    ct = np.maximum(np.cos(theta), 0.0)
    expM = np.exp(c[4] * gamma)
    rayM = cg * cg
    mieM = (1.0 + cg * cg) / ((1.0 + c[8] * c[8] - 2.0 * c[8] * cg) ** 1.5)
    zenith = np.sqrt(ct)
    return (1.0 + c[0] * np.exp(c[1] / (ct + 0.01))) * (c[2] + c[3] * expM + c[5] * rayM + c[6] * mieM + c[7] * zenith)


class HosekState:
    def __init__(self, turbidity, albedo, solar_elevation):
        cfgs, rads = _datasets()
        self.configs = [hosek_cook_config(cfgs[b], turbidity, albedo, solar_elevation) for b in range(11)]
        self.radiances = [hosek_cook_radiance(rads[b], turbidity, albedo, solar_elevation) for b in range(11)]


def hosek_spectral_radiance(state, theta, gamma, wavelength):
    low = int(np.floor((wavelength - 320.0) / 40.0))
    if low < 0 or low >= 11:
        return np.zeros_like(theta)
    interp = ((wavelength - 320.0) / 40.0) % 1.0
    val_low = hosek_radiance(state.configs[low], theta, gamma) * state.radiances[low]
    if interp < 1e-6:
        return val_low
    res = (1.0 - interp) * val_low
    if low + 1 < 11:
        res = res + interp * hosek_radiance(state.configs[low + 1], theta, gamma) * state.radiances[low + 1]
    return res


def _equal_area_square_to_sphere_f32(u, v):
    """equal_area_square_to_sphere (environment_map.jl:131-160) in Float32, as the bake calls it."""
    u = f32(2) * u - f32(1)
    v = f32(2) * v - f32(1)
    up, vp = np.abs(u), np.abs(v)
    sd = f32(1) - (up + vp)
    r = f32(1) - np.abs(sd)
    with np.errstate(invalid="ignore", divide="ignore"):
        phi = np.where(r == 0, f32(1), (vp - up) / np.where(r == 0, f32(1), r) + f32(1)) * f32(np.pi) / f32(4)
    z = np.copysign(f32(1) - r * r, sd)
    rc = r * np.sqrt(f32(2) - r * r)
    return (np.copysign(np.cos(phi), u) * rc).astype(f32), (np.copysign(np.sin(phi), v) * rc).astype(f32), z.astype(f32)


def sunsky_to_envlight(direction, intensity=1.0, turbidity=2.5, ground_albedo=None, ground_enabled=True, resolution=512):
    """-> (EnvironmentLight, SunLight), sun_sky.jl:358-434.  `direction` points TO the sun; z is up."""
    ground_albedo = ground_albedo if ground_albedo is not None else RGBSpectrum(0.3)
    d = np.asarray(direction, dtype=f32)
    d = (f32(1) / np.sqrt((d * d).sum(dtype=f32))) * d
    solar_elevation = float(np.arcsin(np.clip(d[2], f32(0), f32(1))))
    state = HosekState(float(f32(turbidity)), 0.5, solar_elevation)
    n_lambda = 1 + (720 - 320) // 32
    lambdas = np.array([320.0 + i * (720.0 - 320.0) / (n_lambda - 1) for i in range(n_lambda)])
    c = ((np.arange(1, resolution + 1, dtype=f32) - f32(0.5)) / f32(resolution)).astype(f32)
    uu, vv = np.meshgrid(c, c)                                   # sky_data[v_idx, u_idx]
    wx, wy, wz = _equal_area_square_to_sphere_f32(uu, vv)
    theta = np.arccos(np.clip(wz, f32(0), f32(1))).astype(np.float64)
    cos_gamma = np.clip(wx * d[0] + wy * d[1] + wz * d[2], f32(-1), f32(1))
    gamma = np.arccos(cos_gamma).astype(np.float64)
    sky = np.stack([hosek_spectral_radiance(state, theta, gamma, l) for l in lambdas], axis=-1)    # [res, res, 13]
    # _spectrum_to_xyz: piecewise-linear spectrum against the CIE CMFs at 360..830 nm, / CIE_Y_INTEGRAL
    T = _tables.load()
    cie = np.stack(T["cie"]).astype(np.float64)                  # [3, 471]
    lam_cie = 360.0 + np.arange(471)
    idx = np.clip(np.searchsorted(lambdas, lam_cie, side="right") - 1, 0, n_lambda - 2)
    tt = (lam_cie - lambdas[idx]) / (lambdas[idx + 1] - lambdas[idx])
    w_lo = np.where(lam_cie <= lambdas[0], 1.0, np.where(lam_cie >= lambdas[-1], 0.0, 1.0 - tt))
    w_hi = np.where(lam_cie <= lambdas[0], 0.0, np.where(lam_cie >= lambdas[-1], 1.0, tt))
    idx_lo = np.where(lam_cie <= lambdas[0], 0, np.where(lam_cie >= lambdas[-1], n_lambda - 2, idx))
    W = np.zeros((471, n_lambda))
    W[np.arange(471), idx_lo] += w_lo
    W[np.arange(471), idx_lo + 1] += w_hi
    M = (cie @ W) / float(f32(106.856895))                       # [3, 13]
    xyz = (sky @ M.T).astype(f32)                                # [res, res, 3]
    X, Y, Z = xyz[..., 0], xyz[..., 1], xyz[..., 2]
    r = f32(3.2404542) * X - f32(1.5371385) * Y - f32(0.4985314) * Z
    g = f32(-0.9692660) * X + f32(1.8760108) * Y + f32(0.0415560) * Z
    b = f32(0.0556434) * X - f32(0.2040259) * Y + f32(1.0572252) * Z
    data = np.stack([np.maximum(f32(0), r), np.maximum(f32(0), g), np.maximum(f32(0), b)], axis=-1).astype(f32)
    if ground_enabled:
        below = wz <= 0
        data[below] = (np.asarray(ground_albedo.c[:3], dtype=f32) * f32(0.3))
    env = EnvironmentMap(data)
    inv = f32(intensity) / f32(D65_PHOTOMETRIC)
    env_light = EnvironmentLight(env, RGBSpectrum(float(inv)))
    sun_scale = f32(5) * f32(intensity)
    sun = SunLight.from_rgb((float(sun_scale), float(sun_scale * f32(0.95)), float(sun_scale * f32(0.85))), tuple(float(-x) for x in d))
    return env_light, sun

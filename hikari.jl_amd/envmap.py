"""EnvironmentMap + Distribution2D + EnvironmentLight, host side (textures/environment_map.jl:9-66, 78-229;
sampler/sampling.jl:179-262; lights/environment.jl:5-37).

Everything here is scene construction: the tables are built once in Float32, in the reference's accumulation order,
and handed to the C-ABI as an `hk_envmap` record; the device only reads them."""
import numpy as np

from . import _abi as A
from .lights import Light
from .materials import RGBSpectrum

f32 = np.float32


def rotation_matrix(angle_degrees, axis):
    """rotation_matrix(angle, axis) (environment_map.jl:52-66).  Julia's Mat3f(...) constructor is column-major, so the
    nine values written row by row in the reference source fill COLUMNS; the result is returned as M[i][j]."""
    th = f32(np.deg2rad(angle_degrees))
    a = np.asarray(axis, dtype=f32)
    a = (f32(1) / np.sqrt((a * a).sum(dtype=f32))) * a
    s, c = f32(np.sin(th)), f32(np.cos(th))
    t = f32(1) - c
    vals = [t * a[0] * a[0] + c, t * a[0] * a[1] - s * a[2], t * a[0] * a[2] + s * a[1],
            t * a[0] * a[1] + s * a[2], t * a[1] * a[1] + c, t * a[1] * a[2] - s * a[0],
            t * a[0] * a[2] - s * a[1], t * a[1] * a[2] + s * a[0], t * a[2] * a[2] + c]
    return np.array(vals, dtype=f32).reshape(3, 3).T.copy()   # column-major fill


def equal_area_square_to_sphere(u, v):
    """vectorised equal_area_square_to_sphere (environment_map.jl:131-160), float64; used to AUTHOR synthetic maps."""
    u = 2.0 * np.asarray(u, np.float64) - 1.0
    v = 2.0 * np.asarray(v, np.float64) - 1.0
    up, vp = np.abs(u), np.abs(v)
    sd = 1.0 - (up + vp)
    r = 1.0 - np.abs(sd)
    phi = np.where(r == 0, 1.0, (vp - up) / np.where(r == 0, 1.0, r) + 1.0) * np.pi / 4.0
    z = np.copysign(1.0 - r * r, sd)
    rc = r * np.sqrt(2.0 - r * r)
    return np.copysign(np.cos(phi), u) * rc, np.copysign(np.sin(phi), v) * rc, z


class Distribution2D:
    """Distribution2D(func::Matrix{Float32}) (sampling.jl:207-262): piecewise-constant 2-D density, flat storage."""

    def __init__(self, func):
        func = np.ascontiguousarray(func, dtype=f32)
        nv, nu = func.shape
        self.nu, self.nv = nu, nv
        self.conditional_func = func.copy()                              # [v][u] == Julia [nu, nv] column-major
        cdf = np.zeros((nv, nu + 1), dtype=f32)
        np.cumsum(func / f32(nu), axis=1, dtype=f32, out=cdf[:, 1:])      # sequential Float32 adds, like the reference loop
        fint = cdf[:, nu].copy()
        zero = fint == 0                                                 # isapprox(x, 0f0) has atol = 0: exact zero only
        with np.errstate(invalid="ignore", divide="ignore"):
            cdf[:, 1:] = np.where(zero[:, None], (np.arange(1, nu + 1, dtype=f32) / f32(nu))[None, :], cdf[:, 1:] / fint[:, None])
        self.conditional_cdf = cdf
        self.conditional_func_int = fint
        self.marginal_func = fint.copy()
        mc = np.zeros(nv + 1, dtype=f32)
        np.cumsum(fint / f32(nv), dtype=f32, out=mc[1:])
        self.marginal_func_int = f32(mc[nv])
        if self.marginal_func_int == 0:
            mc[1:] = np.arange(1, nv + 1, dtype=f32) / f32(nv)
        else:
            mc[1:] = mc[1:] / self.marginal_func_int
        self.marginal_cdf = mc


class EnvironmentMap:
    """EnvironmentMap(data::Matrix{RGBSpectrum}, rotation) (environment_map.jl:24-45): square equal-area image,
    data[v, u]; the sampling distribution is the texel luminance to_Y (no sin(theta): equal-area)."""

    def __init__(self, data, rotation=None):
        d = np.asarray(data, dtype=f32)
        assert d.ndim == 3 and d.shape[2] in (3, 4)
        if d.shape[2] == 3:
            d = np.concatenate([d, np.ones(d.shape[:2] + (1,), f32)], axis=2)
        self.data = np.ascontiguousarray(d)
        self.height, self.width = d.shape[:2]
        self.rotation = np.eye(3, dtype=f32) if rotation is None else np.asarray(rotation, dtype=f32).reshape(3, 3)
        lum = f32(0.212671) * d[..., 0] + f32(0.715160) * d[..., 1] + f32(0.072169) * d[..., 2]   # to_Y (spectrum.jl:74-76)
        self.distribution = Distribution2D(lum.astype(f32))
        self._jl = np.ascontiguousarray(np.transpose(self.data, (1, 0, 2)))   # [x][y][4] == Julia [h, w] column-major

    def record(self):
        r = A.hk_envmap()
        D = self.distribution
        r.width, r.height = self.width, self.height
        r.data = self._jl.ctypes.data_as(A.PF)
        r.rotation[:] = [float(x) for x in self.rotation.reshape(-1)]
        r.nu, r.nv = D.nu, D.nv
        r.conditional_func = D.conditional_func.ctypes.data_as(A.PF)
        r.conditional_cdf = D.conditional_cdf.ctypes.data_as(A.PF)
        r.conditional_func_int = D.conditional_func_int.ctypes.data_as(A.PF)
        r.marginal_func = D.marginal_func.ctypes.data_as(A.PF)
        r.marginal_cdf = D.marginal_cdf.ctypes.data_as(A.PF)
        r.marginal_func_int = float(D.marginal_func_int)
        return r


class EnvironmentLight(Light):
    """EnvironmentLight(env_map, scale::RGBSpectrum = RGBSpectrum(1)) (lights/environment.jl:5-19); infinite light."""
    kind = A.HK_LIGHT_ENVIRONMENT

    def __init__(self, env_map, scale=None):
        self.env_map = env_map
        self.scale_rgb = scale if scale is not None else RGBSpectrum(1.0)
        self.i = self.scale_rgb
        self.scale = 1.0


def analytic_sky(res=64, sun_dir=(1.0, 2.0, 9.0), sun_radiance=(40.0, 36.0, 30.0), sun_cos=0.995, zenith=(0.15, 0.3, 0.8),
                 horizon=(0.7, 0.75, 0.8), ground=(0.1, 0.09, 0.08)):
    """A synthetic z-up sky authored directly in the equal-area parameterisation: gradient + a small bright sun cap.
    A stand-in for the Hosek-Wilkie bake (lights/sun_sky.jl) with the same structure: smooth sky + tiny hot spot."""
    c = (np.arange(res, dtype=np.float64) + 0.5) / res
    uu, vv = np.meshgrid(c, c)                       # data[v, u]
    x, y, z = equal_area_square_to_sphere(uu, vv)
    s = np.asarray(sun_dir, np.float64)
    s = s / np.linalg.norm(s)
    t = np.clip(z, 0, 1)[..., None] ** 0.5
    sky = np.asarray(horizon)[None, None, :] * (1 - t) + np.asarray(zenith)[None, None, :] * t
    img = np.where((z >= 0)[..., None], sky, np.asarray(ground)[None, None, :])
    cs = x * s[0] + y * s[1] + z * s[2]
    img = img + np.asarray(sun_radiance)[None, None, :] * (cs > sun_cos)[..., None]
    glow = np.clip((cs - 0.9) / 0.1, 0, 1)[..., None] ** 3
    img = img + 0.5 * np.asarray(sun_radiance)[None, None, :] * 0.05 * glow
    return img.astype(f32)

"""Independent float64 references for the oracle (test infrastructure).

Everything here is written from the FORMULAS the reference's Julia text states (and the pbrt-v4 definitions it cites), in
vectorised numpy float64 with numpy's own complex arithmetic / trigonometry — not a transliteration of oracle/*.h or of the HIP
code, and sharing no code with either.  tests/test_independent_pins.py checks the oracle against these; agreement is expected
to float32 rounding (1e-5 relative) because the oracle evaluates the same mathematics in binary32.

    fresnel_dielectric      materials/spectral-eval.jl (fresnel_dielectric); pbrt-v4 FrDielectric
    fr_complex              spectral-eval.jl:3663-3754; pbrt-v4 FrComplex, here with numpy complex sqrt
    tr_d / tr_lambda / ...  spectral-eval.jl:3765-3864; pbrt-v4 TrowbridgeReitzDistribution
    hg_phase / sample_hg    integrators/volpath/media.jl:16-76
    equal_area_*            textures/environment_map.jl:78-160 (Clarberg's mapping), here through arctan2
    PiecewiseConstant2D     sampler/sampling.jl:179-361
    node_importance         lights/bvh-light-sampler.jl:58-91
"""
import numpy as np

PI = np.pi


def _f64(x):
    return np.asarray(x, dtype=np.float64)


# ---------------------------------------------------------------------------------------------------- Fresnel
def fresnel_dielectric(cos_i, eta):
    """Unpolarised Fresnel reflectance of a dielectric interface, eta = n_t / n_i; a negative cosine means the ray leaves the
    medium (eta -> 1/eta).  Total internal reflection -> 1."""
    c = np.clip(_f64(cos_i), -1.0, 1.0)
    eta = np.broadcast_to(_f64(eta), c.shape).copy()
    flip = c < 0
    eta[flip] = 1.0 / eta[flip]
    c = np.abs(c)
    s2t = (1.0 - c * c) / (eta * eta)
    tir = s2t >= 1.0
    ct = np.sqrt(np.maximum(0.0, 1.0 - s2t))
    r_parl = (eta * c - ct) / (eta * c + ct)
    r_perp = (c - eta * ct) / (c + eta * ct)
    return np.where(tir, 1.0, 0.5 * (r_parl ** 2 + r_perp ** 2))


def fr_complex(cos_i, eta, k):
    """Conductor Fresnel with complex index eta + i k: numpy complex arithmetic (principal square root)."""
    c = np.clip(_f64(cos_i), 0.0, 1.0)
    n = _f64(eta) + 1j * _f64(k)
    s2i = 1.0 - c * c
    s2t = s2i / (n * n)
    ct = np.sqrt(1.0 - s2t)
    r_parl = (n * c - ct) / (n * c + ct)
    r_perp = (c - n * ct) / (c + n * ct)
    return 0.5 * (np.abs(r_parl) ** 2 + np.abs(r_perp) ** 2)


# ---------------------------------------------------------------------------------------------------- Trowbridge-Reitz
def _angles(w):
    w = _f64(w)
    cos_t = w[..., 2]
    sin2 = np.maximum(0.0, 1.0 - cos_t ** 2)
    sin_t = np.sqrt(sin2)
    with np.errstate(divide="ignore", invalid="ignore"):
        cos_p = np.where(sin_t == 0, 1.0, np.clip(w[..., 0] / sin_t, -1, 1))
        sin_p = np.where(sin_t == 0, 0.0, np.clip(w[..., 1] / sin_t, -1, 1))
        tan2 = sin2 / cos_t ** 2
    return cos_t, tan2, cos_p, sin_p


def tr_d(wm, ax, ay):
    """D(wm) = 1 / (pi ax ay cos^4 (1 + tan^2 (cos^2phi/ax^2 + sin^2phi/ay^2))^2)"""
    cos_t, tan2, cp, sp = _angles(wm)
    cos4 = cos_t ** 4
    e = tan2 * ((cp / ax) ** 2 + (sp / ay) ** 2)
    with np.errstate(divide="ignore", invalid="ignore"):
        d = 1.0 / (PI * ax * ay * cos4 * (1.0 + e) ** 2)
    return np.where(np.isinf(tan2) | (cos4 < 1e-16), 0.0, d)


def tr_lambda(w, ax, ay):
    cos_t, tan2, cp, sp = _angles(w)
    a2 = (cp * ax) ** 2 + (sp * ay) ** 2
    with np.errstate(invalid="ignore"):
        lam = (np.sqrt(1.0 + a2 * tan2) - 1.0) / 2.0
    return np.where(np.isinf(tan2), 0.0, lam)


def tr_g1(w, ax, ay):
    return 1.0 / (1.0 + tr_lambda(w, ax, ay))


def tr_g(wo, wi, ax, ay):
    return 1.0 / (1.0 + tr_lambda(wo, ax, ay) + tr_lambda(wi, ax, ay))


def tr_pdf(w, wm, ax, ay):
    """visible-normal density D_w(wm) = G1(w) / |cos w| * D(wm) * |w . wm|"""
    w, wm = _f64(w), _f64(wm)
    return tr_g1(w, ax, ay) / np.abs(w[..., 2]) * tr_d(wm, ax, ay) * np.abs((w * wm).sum(-1))


def _normalize(v):
    return v / np.linalg.norm(v, axis=-1, keepdims=True)


def tr_sample_wm(w, u, ax, ay):
    """pbrt-v4 Sample_wm: stretch, orthonormal frame around the stretched direction, uniform disk warped by the hemisphere
    projection, unstretch."""
    w, u = _f64(w), _f64(u)
    wh = _normalize(np.stack([ax * w[..., 0], ay * w[..., 1], w[..., 2]], -1))
    wh = np.where(wh[..., 2:3] < 0, -wh, wh)
    z = np.zeros_like(wh)
    z[..., 2] = 1.0
    t1 = np.where(wh[..., 2:3] < 0.99999, _normalize(np.cross(z, wh) + (wh[..., 2:3] >= 0.99999) * np.array([1.0, 0, 0])), np.array([1.0, 0, 0]))
    t2 = np.cross(wh, t1)
    r = np.sqrt(u[..., 0])
    phi = 2 * PI * u[..., 1]
    px, py = r * np.cos(phi), r * np.sin(phi)
    h = np.sqrt(1 - px * px)
    t = 0.5 * (1 + wh[..., 2])
    py = (1 - t) * h + t * py
    pz = np.sqrt(np.maximum(0.0, 1 - px * px - py * py))
    nh = px[..., None] * t1 + py[..., None] * t2 + pz[..., None] * wh
    return _normalize(np.stack([ax * nh[..., 0], ay * nh[..., 1], np.maximum(1e-6, nh[..., 2])], -1))


def conductor_f(wo, wi, ax, ay, eta, k):
    """rough conductor BRDF D F G / (4 cos_i cos_o) and its sampling pdf D_wo(wm) / (4 |wo.wm|), local frame, same hemisphere"""
    wo, wi = _f64(wo), _f64(wi)
    wm = _normalize(wo + wi)
    F = fr_complex(np.abs((wo * wm).sum(-1))[..., None], eta, k)
    f = (tr_d(wm, ax, ay) * tr_g(wo, wi, ax, ay) / (4 * np.abs(wi[..., 2]) * np.abs(wo[..., 2])))[..., None] * F
    wm_f = np.where(wm[..., 2:3] < 0, -wm, wm)
    pdf = tr_pdf(wo, wm_f, ax, ay) / (4 * np.abs((wo * wm_f).sum(-1)))
    return f, pdf


# ---------------------------------------------------------------------------------------------------- frames / sampling
def coordinate_system(n):
    """spectral-eval.jl:3514-3533 (the older pbrt-v3 construction, NOT Frisvad/Duff)"""
    n = _f64(n)
    a = np.abs(n[..., 0]) > np.abs(n[..., 1])
    with np.errstate(divide="ignore", invalid="ignore"):
        l1 = 1.0 / np.sqrt(n[..., 0] ** 2 + n[..., 2] ** 2)
        l2 = 1.0 / np.sqrt(n[..., 1] ** 2 + n[..., 2] ** 2)
    t_a = np.stack([n[..., 2] * l1, np.zeros_like(l1), -n[..., 0] * l1], -1)
    t_b = np.stack([np.zeros_like(l2), n[..., 2] * l2, -n[..., 1] * l2], -1)
    t = np.where(a[..., None], t_a, t_b)
    return t, np.cross(n, t)


def concentric_disk(u):
    u = _f64(u)
    o = 2 * u - 1
    ox, oy = o[..., 0], o[..., 1]
    with np.errstate(divide="ignore", invalid="ignore"):
        a = np.abs(ox) > np.abs(oy)
        r = np.where(a, ox, oy)
        th = np.where(a, (PI / 4) * (oy / ox), PI / 2 - (PI / 4) * (ox / oy))
    zero = (ox == 0) & (oy == 0)
    return np.where(zero, 0.0, r * np.cos(th)), np.where(zero, 0.0, r * np.sin(th))


def cosine_hemisphere(u):
    x, y = concentric_disk(u)
    return np.stack([x, y, np.sqrt(np.maximum(0.0, 1 - x * x - y * y))], -1)


# ---------------------------------------------------------------------------------------------------- Henyey-Greenstein
def hg_phase(g, cos_t):
    """p(cos) = (1 - g^2) / (4 pi (1 + g^2 - 2 g cos)^(3/2))   (media.jl:31-40: the sign Hikari's text states)"""
    d = 1 + g * g - 2 * g * _f64(cos_t)
    return (1 - g * g) / (4 * PI * d * np.sqrt(np.maximum(d, 0)))


def hg_cos_from_u(g, u0):
    """inverse CDF of the HG polar angle measured from -wo (media.jl:51-63): isotropic for |g| < 1e-3"""
    u0 = _f64(u0)
    if abs(g) < 1e-3:
        return 1 - 2 * u0
    return np.clip((1 + g * g - ((1 - g * g) / (1 - g + 2 * g * u0)) ** 2) / (2 * g), -1, 1)


def hg_cdf(g, c):
    """P(cos <= c) under the density 2 pi hg_phase(g, cos): closed form of the integral of (1-g^2)/(2 (1+g^2-2gx)^(3/2)) dx"""
    c = _f64(c)
    if abs(g) < 1e-3:
        return (c + 1) / 2
    return (1 - g * g) / (2 * g) * (1 / np.sqrt(1 + g * g - 2 * g * c) - 1 / (1 + g))


# ---------------------------------------------------------------------------------------------------- equal-area mapping
def equal_area_square_to_sphere(p):
    """Clarberg's octahedral equal-area map [0,1]^2 -> S^2 (z up), through exact trigonometry."""
    p = _f64(p)
    u, v = 2 * p[..., 0] - 1, 2 * p[..., 1] - 1
    up, vp = np.abs(u), np.abs(v)
    sd = 1 - (up + vp)
    d = np.abs(sd)
    r = 1 - d
    with np.errstate(divide="ignore", invalid="ignore"):
        phi = np.where(r == 0, 1.0, (vp - up) / r + 1) * PI / 4
    z = np.copysign(1 - r * r, sd)
    s = r * np.sqrt(np.maximum(0.0, 2 - r * r))
    return np.stack([np.copysign(np.cos(phi), u) * s, np.copysign(np.sin(phi), v) * s, z], -1)


def equal_area_sphere_to_square(d):
    d = _f64(d)
    x, y, z = np.abs(d[..., 0]), np.abs(d[..., 1]), np.abs(d[..., 2])
    r = np.sqrt(np.maximum(0.0, 1 - z))
    a, b = np.maximum(x, y), np.minimum(x, y)
    with np.errstate(divide="ignore", invalid="ignore"):
        phi = np.where(a == 0, 0.0, np.arctan2(b, a)) * 2 / PI
    phi = np.where(x < y, 1 - phi, phi)
    v = phi * r
    u = r - v
    neg = d[..., 2] < 0
    u, v = np.where(neg, 1 - v, u), np.where(neg, 1 - u, v)
    u, v = np.copysign(u, d[..., 0]), np.copysign(v, d[..., 1])
    return np.stack([(u + 1) / 2, (v + 1) / 2], -1)


# ---------------------------------------------------------------------------------------------------- Distribution2D
class PiecewiseConstant2D:
    """func[nv][nu] (row = v); continuous sampling: marginal over v from u[1], conditional over u from u[0];
    pdf(u, v) = func[iv][iu] / integral  (sampling.jl:179-361)"""

    def __init__(self, func):
        f = _f64(func)
        self.f = f
        self.nv, self.nu = f.shape
        self.row_int = f.mean(axis=1)                       # integral of each row over u in [0,1]
        self.total = self.row_int.mean()
        self.row_cdf = np.concatenate([np.zeros((self.nv, 1)), np.cumsum(f, axis=1) / self.nu], axis=1)
        with np.errstate(divide="ignore", invalid="ignore"):
            self.row_cdf = np.where(self.row_int[:, None] > 0, self.row_cdf / self.row_int[:, None], np.linspace(0, 1, self.nu + 1)[None, :])
        m = np.concatenate([[0.0], np.cumsum(self.row_int) / self.nv])
        self.marg_cdf = m / m[-1] if m[-1] > 0 else np.linspace(0, 1, self.nv + 1)

    @staticmethod
    def _sample_1d(cdf, func, integral, u):
        n = len(func)
        i = int(np.clip(np.searchsorted(cdf, u, side="right") - 1, 0, n - 1))
        du = u - cdf[i]
        if cdf[i + 1] - cdf[i] > 0:
            du /= cdf[i + 1] - cdf[i]
        return (i + du) / n, (func[i] / integral if integral > 0 else 0.0), i

    def sample(self, u):
        v, pv, iv = self._sample_1d(self.marg_cdf, self.row_int, self.total, u[1])
        uu, pu, iu = self._sample_1d(self.row_cdf[iv], self.f[iv], self.row_int[iv], u[0])
        return (uu, v), pu * pv

    def pdf(self, uv):
        iu = int(np.clip(int(uv[0] * self.nu), 0, self.nu - 1))
        iv = int(np.clip(int(uv[1] * self.nv), 0, self.nv - 1))
        return self.f[iv, iu] / self.total


# ---------------------------------------------------------------------------------------------------- light BVH importance
def node_importance(p, n, bmin, bmax, w, phi, cos_o, cos_e, two_sided):
    """pbrt-v4 LightBounds::Importance as Hikari states it (bvh-light-sampler.jl:58-91): note d2 = max(d2, |diag| / 2) — the
    LENGTH of the half diagonal, not its square (quirk Q15)."""
    p, n, bmin, bmax, w = (_f64(x) for x in (p, n, bmin, bmax, w))
    pc = (bmin + bmax) / 2
    d2 = ((p - pc) ** 2).sum()
    d2 = max(d2, np.linalg.norm(bmax - bmin) / 2)
    wi = (p - pc) / np.linalg.norm(p - pc)
    cos_w = float(np.dot(w, wi))
    if two_sided:
        cos_w = abs(cos_w)
    sin_w = np.sqrt(max(0.0, 1 - cos_w * cos_w))
    # cosine of the angle subtended by the bounding sphere of the box, seen from p
    r2 = ((bmax - pc) ** 2).sum()
    if ((p - pc) ** 2).sum() < r2:        # inside the bounding SPHERE -> whole sphere of directions (light-bounds.jl:96-109)
        cos_b = -1.0
    else:
        cos_b = np.sqrt(max(0.0, 1 - r2 / ((p - pc) ** 2).sum()))
    sin_b = np.sqrt(max(0.0, 1 - cos_b * cos_b))
    sin_o = np.sqrt(max(0.0, 1 - cos_o * cos_o))

    def cos_sub(sa, ca, sb, cb):       # cos(max(0, a - b))
        return 1.0 if ca > cb else ca * cb + sa * sb

    def sin_sub(sa, ca, sb, cb):
        return 0.0 if ca > cb else sa * cb - ca * sb

    cos_x = cos_sub(sin_w, cos_w, sin_o, cos_o)
    sin_x = sin_sub(sin_w, cos_w, sin_o, cos_o)
    cos_p = cos_sub(sin_x, cos_x, sin_b, cos_b)
    if cos_p <= cos_e:
        return 0.0
    imp = phi * cos_p / d2
    if not (n[0] == 0 and n[1] == 0 and n[2] == 0):
        cos_i = abs(float(np.dot(wi, n)))
        sin_i = np.sqrt(max(0.0, 1 - cos_i * cos_i))
        cos_pi = cos_sub(sin_i, cos_i, sin_b, cos_b)
        imp *= cos_pi
    return max(imp, 0.0)

"""CoatedDiffuseMaterial — pbrt-v4's LayeredBxDF as the reference ports it — restated in scalar float32 NumPy from the Julia text
(materials/spectral-eval.jl:815-1940): the dielectric top interface (smooth or Trowbridge-Reitz rough: sample / evaluate / pdf), the
Lambertian base, the random walk of sample_bsdf_spectral (:1232-1414), the nSamples walks with next-event estimation and MIS of
evaluate_bsdf_spectral (:1564-1836) and the pdf estimate (:1848-1928).  Test infrastructure (a second source beside oracle/hko_layered.h);
shares nothing with it.  The walks draw from a PCG32 seeded by MurmurHash64A of the float BITS of (wo), (uc, u) or (wi): a comparison is
exact only on identical inputs, which is how tests/test_layered_pin.py uses it."""
import struct

import numpy as np

from ref_volpath_np import PCG32, PI, coordinate_system, cosine_hemisphere, f32, murmur64a

EPS = f32(np.finfo(np.float32).eps)
REFL, TRANS, ALL = 1, 2, 3


def V(x, y, z):
    return np.array([x, y, z], np.float32)


def _r(fn, x):
    """a transcendental of a binary32 number rounded once (through binary64)"""
    return f32(fn(np.float64(f32(x))))


def dot(a, b):
    return f32(f32(f32(a[0] * b[0]) + f32(a[1] * b[1])) + f32(a[2] * b[2]))


def normalize(v):
    with np.errstate(divide="ignore", invalid="ignore"):
        return (v / np.sqrt(dot(v, v))).astype(np.float32)


def cross(a, b):
    return V(a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0])


def hash_seed_vec(v):           # pbrt_hash(seed = 0 :: UInt64, v :: Vec3f)   spectral-eval.jl:718-726
    return murmur64a(struct.pack("<Q3f", 0, float(v[0]), float(v[1]), float(v[2])), 0)


def hash_vec(v):                # pbrt_hash(v :: Vec3f)                       :700-707
    return murmur64a(struct.pack("<3f", float(v[0]), float(v[1]), float(v[2])), 0)


def hash_f_p2(a, b):            # pbrt_hash(a :: Float32, b :: Point2f)        :735-741
    return murmur64a(struct.pack("<3f", float(a), float(b[0]), float(b[1])), 0)


# ---------------------------------------------------------------------------------------------------- trigonometry of local directions
def cos2(w):
    return f32(w[2] * w[2])


def sin2(w):
    return max(f32(0), f32(f32(1) - cos2(w)))


def tan2(w):
    with np.errstate(divide="ignore", invalid="ignore"):
        return f32(sin2(w) / cos2(w))


def cos_phi(w):
    s = f32(np.sqrt(sin2(w)))
    return f32(1) if s == 0 else min(max(f32(w[0] / s), f32(-1)), f32(1))


def sin_phi(w):
    s = f32(np.sqrt(sin2(w)))
    return f32(0) if s == 0 else min(max(f32(w[1] / s), f32(-1)), f32(1))


def same_hemisphere(a, b):
    return f32(a[2] * b[2]) > 0


# ---------------------------------------------------------------------------------------------------- Trowbridge-Reitz (:3774-3861)
def tr_smooth(ax, ay):
    return max(ax, ay) < f32(1e-3)


def tr_d(wm, ax, ay):
    t2 = tan2(wm)
    if np.isinf(t2):
        return f32(0)
    c4 = f32(cos2(wm) * cos2(wm))
    if c4 < f32(1e-16):
        return f32(0)
    a, b = f32(cos_phi(wm) / ax), f32(sin_phi(wm) / ay)
    e = f32(t2 * f32(f32(a * a) + f32(b * b)))
    return f32(f32(1) / f32(f32(f32(f32(PI * ax) * ay) * c4) * f32(f32(f32(1) + e) * f32(f32(1) + e))))


def tr_lambda(w, ax, ay):
    t2 = tan2(w)
    if np.isinf(t2):
        return f32(0)
    a, b = f32(cos_phi(w) * ax), f32(sin_phi(w) * ay)
    a2 = f32(f32(a * a) + f32(b * b))
    return f32(f32(np.sqrt(f32(f32(1) + f32(a2 * t2))) - f32(1)) * f32(0.5))


def tr_g1(w, ax, ay):
    return f32(f32(1) / f32(f32(1) + tr_lambda(w, ax, ay)))


def tr_g(wo, wi, ax, ay):
    return f32(f32(1) / f32(f32(f32(1) + tr_lambda(wo, ax, ay)) + tr_lambda(wi, ax, ay)))


def tr_pdf(w, wm, ax, ay):
    with np.errstate(divide="ignore", invalid="ignore"):
        return f32(f32(f32(tr_g1(w, ax, ay) / abs(w[2])) * tr_d(wm, ax, ay)) * abs(dot(w, wm)))


def tr_sample_wm(w, u, ax, ay):
    wh = normalize(V(ax * w[0], ay * w[1], w[2]))
    if wh[2] < 0:
        wh = -wh
    t1 = normalize(cross(V(0, 0, 1), wh)) if wh[2] < f32(0.99999) else V(1, 0, 0)
    t2 = cross(wh, t1)
    r = f32(np.sqrt(u[0]))
    phi = f32(f32(f32(2) * PI) * u[1])
    px, py = f32(r * _r(np.cos, phi)), f32(r * _r(np.sin, phi))
    h = f32(np.sqrt(f32(f32(1) - f32(px * px))))
    t = f32(f32(0.5) * f32(f32(1) + wh[2]))                    # lerp(h, p_y, t) = (1 - t) h + t p_y
    py = f32(f32(f32(f32(1) - t) * h) + f32(t * py))
    pz = f32(np.sqrt(max(f32(0), f32(f32(f32(1) - f32(px * px)) - f32(py * py)))))
    nh = (px * t1 + py * t2 + pz * wh).astype(np.float32)
    return normalize(V(ax * nh[0], ay * nh[1], max(f32(1e-6), nh[2])))


# ---------------------------------------------------------------------------------------------------- Fresnel, refraction (reflection/bxdf.jl:67-90; :1072-1125)
def fresnel(cos_i, eta):
    c = min(max(f32(cos_i), f32(-1)), f32(1))
    eta = f32(eta)
    if c < 0:
        eta, c = f32(f32(1) / eta), f32(-c)
    s2t = f32(f32(f32(1) - f32(c * c)) / f32(eta * eta))
    if s2t >= 1:
        return f32(1)
    ct = f32(np.sqrt(f32(f32(1) - s2t)))
    rp = f32(f32(f32(eta * c) - ct) / f32(f32(eta * c) + ct))
    rs = f32(f32(c - f32(eta * ct)) / f32(c + f32(eta * ct)))
    return f32(f32(0.5) * f32(f32(rp * rp) + f32(rs * rs)))


def refract_pbrt(wo, eta):
    ci = wo[2]
    etap = f32(eta) if ci > 0 else f32(f32(1) / f32(eta))
    s2i = max(f32(0), f32(f32(1) - f32(ci * ci)))
    s2t = f32(s2i / f32(etap * etap))
    if s2t >= 1:
        return False, None, f32(1)
    ct = f32(np.sqrt(f32(f32(1) - s2t)))
    return True, normalize(V(f32(-wo[0] / etap), f32(-wo[1] / etap), -ct if ci > 0 else ct)), etap


def refract_microfacet(wo, wm, eta):
    ci = dot(wo, wm)
    etap = f32(eta) if ci > 0 else f32(f32(1) / f32(eta))
    s2i = max(f32(0), f32(f32(1) - f32(ci * ci)))
    s2t = f32(s2i / f32(etap * etap))
    if s2t >= 1:
        return False, None, f32(1)
    ct = f32(np.sqrt(f32(f32(1) - s2t)))
    cts = -ct if ci > 0 else ct
    wi = ((-wo / etap).astype(np.float32) + (f32(f32(ci / etap) + cts) * wm).astype(np.float32)).astype(np.float32)
    return True, normalize(wi), etap


# ---------------------------------------------------------------------------------------------------- the two interfaces
class S:      # LayeredBSDFSample
    def __init__(self, f=None, wi=None, pdf=0.0, refl=False, spec=False, eta=1.0, valid=False):
        self.f = np.zeros(4, np.float32) if f is None else np.asarray(f, np.float32)
        self.wi, self.pdf, self.refl, self.spec, self.eta, self.valid = wi, f32(pdf), refl, spec, f32(eta), valid


def s4(x):
    return np.full(4, f32(x), np.float32)


def sample_dielectric(wo, uc, u, ax, ay, eta, flags):
    """sample_dielectric_interface (:973-1066)"""
    with np.errstate(divide="ignore", invalid="ignore"):
        if tr_smooth(ax, ay) or eta == 1:
            R = fresnel(wo[2], eta)
            T = f32(f32(1) - R)
            pr, pt = (R if flags & REFL else f32(0)), (T if flags & TRANS else f32(0))
            if pr == 0 and pt == 0:
                return S()
            if uc < f32(pr / f32(pr + pt)):
                wi = V(-wo[0], -wo[1], wo[2])
                return S(s4(f32(R / abs(wi[2]))), wi, f32(pr / f32(pr + pt)), True, True, 1.0, True)
            ok, wi, etap = refract_pbrt(wo, eta)
            if not ok:
                return S()
            return S(s4(f32(T / abs(wi[2]))), wi, f32(pt / f32(pr + pt)), False, True, etap, True)
        wm = tr_sample_wm(wo, u, ax, ay)
        com = dot(wo, wm)
        R = fresnel(com, eta)
        T = f32(f32(1) - R)
        pr, pt = (R if flags & REFL else f32(0)), (T if flags & TRANS else f32(0))
        if pr == 0 and pt == 0:
            return S()
        if uc < f32(pr / f32(pr + pt)):
            wi = ((-wo).astype(np.float32) + (f32(f32(2) * dot(wo, wm)) * wm).astype(np.float32)).astype(np.float32)
            if not same_hemisphere(wo, wi):
                return S()
            pdf = f32(f32(f32(tr_pdf(wo, wm, ax, ay) / f32(f32(4) * abs(com))) * pr) / f32(pr + pt))
            fv = f32(f32(f32(tr_d(wm, ax, ay) * tr_g(wo, wi, ax, ay)) * R) / f32(f32(f32(4) * wo[2]) * wi[2]))
            return S(s4(fv), wi, pdf, True, False, 1.0, True)
        ok, wi, etap = refract_microfacet(wo, wm, eta)
        if (not ok) or same_hemisphere(wo, wi) or wi[2] == 0:
            return S()
        t = f32(dot(wi, wm) + f32(dot(wo, wm) / etap))
        denom = f32(t * t)
        dwm = f32(abs(dot(wi, wm)) / denom)
        pdf = f32(f32(f32(tr_pdf(wo, wm, ax, ay) * dwm) * pt) / f32(pr + pt))
        fv = f32(f32(f32(T * tr_d(wm, ax, ay)) * tr_g(wo, wi, ax, ay)) * abs(f32(f32(dot(wi, wm) * dot(wo, wm)) / f32(f32(wi[2] * wo[2]) * denom))))
        return S(s4(fv), wi, pdf, False, False, etap, True)


def _half(wo, wi, etap=None):
    wh = normalize((wo + wi).astype(np.float32) if etap is None else (wo + (wi * etap).astype(np.float32)).astype(np.float32))
    return -wh if wh[2] < 0 else wh


def eval_dielectric(wo, wi, ax, ay, eta):
    """eval_dielectric_interface (:1426-1485) -> (f scalar, pdf)"""
    with np.errstate(divide="ignore", invalid="ignore"):
        if tr_smooth(ax, ay) or eta == 1:
            return f32(0), f32(0)
        if same_hemisphere(wo, wi):
            wh = _half(wo, wi)
            coh = dot(wo, wh)
            R = fresnel(coh, eta)
            fv = f32(f32(f32(tr_d(wh, ax, ay) * tr_g(wo, wi, ax, ay)) * R) / f32(f32(f32(4) * wo[2]) * wi[2]))
            return fv, f32(tr_pdf(wo, wh, ax, ay) / f32(f32(4) * abs(coh)))
        etap = f32(eta) if wo[2] > 0 else f32(f32(1) / f32(eta))
        wh = _half(wo, wi, etap)
        coh, cih = dot(wo, wh), dot(wi, wh)
        if f32(coh * cih) > 0:
            return f32(0), f32(0)
        T = f32(f32(1) - fresnel(coh, eta))
        t = f32(cih + f32(coh / etap))
        denom = f32(t * t)
        fv = f32(f32(f32(T * tr_d(wh, ax, ay)) * tr_g(wo, wi, ax, ay)) * abs(f32(f32(cih * coh) / f32(f32(wo[2] * wi[2]) * denom))))
        return fv, f32(tr_pdf(wo, wh, ax, ay) * f32(abs(cih) / denom))


def pdf_dielectric(wo, wi, ax, ay, eta, flags=ALL):
    """pdf_dielectric_interface (:1493-1551)"""
    with np.errstate(divide="ignore", invalid="ignore"):
        if tr_smooth(ax, ay) or eta == 1:
            return f32(0)
        if same_hemisphere(wo, wi):
            if not flags & REFL:
                return f32(0)
            wh = _half(wo, wi)
            coh = abs(dot(wo, wh))
            R = fresnel(coh, eta)
            pr, pt = (R if flags & REFL else f32(0)), (f32(f32(1) - R) if flags & TRANS else f32(0))
            return f32(f32(f32(tr_pdf(wo, wh, ax, ay) / f32(f32(4) * coh)) * pr) / f32(pr + pt))
        if not flags & TRANS:
            return f32(0)
        etap = f32(eta) if wo[2] > 0 else f32(f32(1) / f32(eta))
        wh = _half(wo, wi, etap)
        coh, cih = dot(wo, wh), dot(wi, wh)
        if f32(coh * cih) > 0:
            return f32(0)
        R = fresnel(abs(coh), eta)
        pr, pt = (R if flags & REFL else f32(0)), (f32(f32(1) - R) if flags & TRANS else f32(0))
        t = f32(cih + f32(coh / etap))
        denom = f32(t * t)
        return f32(f32(f32(tr_pdf(wo, wh, ax, ay) * f32(abs(cih) / denom)) * pt) / f32(pr + pt))


def sample_diffuse(wo, u, refl, flags):
    """sample_diffuse_interface (:1144-1170): reflection only"""
    if not flags & REFL:
        return S()
    wi = cosine_hemisphere(np.array([u[0]], np.float32), np.array([u[1]], np.float32))[0]
    if wo[2] < 0:
        wi = V(wi[0], wi[1], -wi[2])
    c = abs(wi[2])
    if c < f32(1e-6):
        return S()
    return S((refl * f32(f32(1) / PI)).astype(np.float32), wi, f32(c / PI), True, False, 1.0, True)


def eval_diffuse(wo, wi, refl):
    if not same_hemisphere(wo, wi):
        return np.zeros(4, np.float32), f32(0)
    return (refl * f32(f32(1) / PI)).astype(np.float32), f32(abs(wi[2]) / PI)


def pdf_diffuse(wo, wi):
    return f32(abs(wi[2]) / PI) if same_hemisphere(wo, wi) else f32(0)


class DiffuseBottom:
    """CoatedDiffuse's base: the Lambertian interface above (sample_diffuse_interface / eval_ / pdf_, :1144-1199)"""

    def __init__(self, refl):
        self.refl = np.asarray(refl, np.float32)

    def sample(self, wo, u, uc, flags):
        return sample_diffuse(wo, u, self.refl, flags)

    def eval(self, wo, wi):
        return eval_diffuse(wo, wi, self.refl)

    def pdf(self, wo, wi):
        return pdf_diffuse(wo, wi)

    # the draws pdf_layered_bsdf takes for its two base samples (:1893-1894, :1912-1913)
    def draw_rs(self, rng):
        return (rng.f32(), rng.f32()), f32(0)

    def draw_wis(self, rng):
        return (rng.f32(), rng.f32()), f32(0)


class DiffuseTransmissionBottom:
    """CoatedDiffuseTransmission's base (sample_ / eval_ / pdf_diffuse_transmission_bottom, :2257-2335): a Lambertian lobe on wo's side with
    probability pr / (pr + pt), one on the other side otherwise (pr, pt: the largest components of the clamped reflectance / transmittance)"""

    def __init__(self, refl, trans, pr, pt):
        self.refl, self.trans, self.pr, self.pt = np.asarray(refl, np.float32), np.asarray(trans, np.float32), f32(pr), f32(pt)

    def sample(self, wo, u, uc, flags):
        pr = self.pr if flags & REFL else f32(0)
        pt = self.pt if flags & TRANS else f32(0)
        if f32(pr + pt) < f32(1e-10):
            return S()
        prob_r = f32(pr / f32(pr + pt))
        wi = cosine_hemisphere(np.array([u[0]], np.float32), np.array([u[1]], np.float32))[0]
        if uc < prob_r:
            if wo[2] < 0:
                wi = V(wi[0], wi[1], -wi[2])
            c = abs(wi[2])
            if c < f32(1e-6):
                return S()
            return S((self.refl * f32(f32(1) / PI)).astype(np.float32), wi, f32(f32(prob_r * c) / PI), True, False, 1.0, True)
        if wo[2] > 0:
            wi = V(wi[0], wi[1], -wi[2])
        c = abs(wi[2])
        if c < f32(1e-6):
            return S()
        return S((self.trans * f32(f32(1) / PI)).astype(np.float32), wi, f32(f32(f32(f32(1) - prob_r) * c) / PI), False, False, 1.0, True)

    def eval(self, wo, wi):
        if f32(self.pr + self.pt) < f32(1e-10):
            return np.zeros(4, np.float32), f32(0)
        c = abs(wi[2])
        if same_hemisphere(wo, wi):
            return (self.refl * f32(f32(1) / PI)).astype(np.float32), f32(f32(f32(self.pr / f32(self.pr + self.pt)) * c) / PI)
        return (self.trans * f32(f32(1) / PI)).astype(np.float32), f32(f32(f32(self.pt / f32(self.pr + self.pt)) * c) / PI)

    def pdf(self, wo, wi):
        if f32(self.pr + self.pt) < f32(1e-10):
            return f32(0)
        c = abs(wi[2])
        share = f32(self.pr / f32(self.pr + self.pt)) if same_hemisphere(wo, wi) else f32(self.pt / f32(self.pr + self.pt))
        return f32(f32(share * c) / PI)

    # pdf_layered_bsdf_dt draws u5, u6, uc3 for its base reflection sample and uc2, u3, u4 for the exit sample (:2790-2792, :2812-2814)
    def draw_rs(self, rng):
        u = (rng.f32(), rng.f32())
        return u, rng.f32()

    def draw_wis(self, rng):
        uc = rng.f32()
        return (rng.f32(), rng.f32()), uc


def power_heuristic(fp, gp):
    f2, g2 = f32(fp * fp), f32(gp * gp)
    return f32(0) if f32(f2 + g2) == 0 else f32(f2 / f32(f2 + g2))


def layer_tr(dz, w):
    if abs(dz) <= EPS:
        return f32(1)
    with np.errstate(divide="ignore", invalid="ignore"):
        return _r(np.exp, -abs(f32(dz / w[2])))


def hg_pdf(g, c):
    g2 = f32(g * g)
    den = f32(f32(f32(1) + g2) - f32(f32(f32(2) * g) * c))
    return f32(f32(f32(1) - g2) / f32(f32(f32(f32(4) * PI) * den) * f32(np.sqrt(max(f32(1e-10), den)))))


def sample_hg(g, wo, u):
    """sample_hg_phase_spectral (:838-868)"""
    if abs(g) < f32(1e-3):
        c = f32(f32(1) - f32(f32(2) * u[0]))
    else:
        g2 = f32(g * g)
        sq = f32(f32(f32(1) - g2) / f32(f32(f32(1) - g) + f32(f32(f32(2) * g) * u[0])))
        c = min(max(f32(f32(f32(f32(1) + g2) - f32(sq * sq)) / f32(f32(2) * g)), f32(-1)), f32(1))
    s = f32(np.sqrt(max(f32(0), f32(f32(1) - f32(c * c)))))
    phi = f32(f32(f32(2) * PI) * u[1])
    mw = (-wo).astype(np.float32)
    t1, t2 = coordinate_system(mw[None])
    t1, t2 = t1[0], t2[0]
    wi = normalize((f32(s * _r(np.cos, phi)) * t1 + f32(s * _r(np.sin, phi)) * t2 + c * mw).astype(np.float32))
    return wi, hg_pdf(g, c)


def mx(v):
    return f32(np.max(v))


# ---------------------------------------------------------------------------------------------------- CoatedDiffuse
class Coated:
    """the evaluated parameters of one CoatedDiffuseMaterial at one wavelength set: refl / albedo [4] (uplifted), alpha_x / alpha_y (after
    roughness_to_alpha when remapped), eta, thickness (>= eps), g (clamped to +-0.99), has_medium, max_depth, n_samples"""

    def __init__(self, refl, albedo, has_medium, ax, ay, eta, thickness, g, max_depth, n_samples, bottom=None):
        self.refl, self.albedo, self.has_medium = np.asarray(refl, np.float32), np.asarray(albedo, np.float32), bool(has_medium)
        self.bottom = bottom if bottom is not None else DiffuseBottom(refl)      # (CoatedDiffuseTransmission: a DiffuseTransmissionBottom — the walks are the same text, :2341-2840)
        self.ax, self.ay, self.eta = f32(ax), f32(ay), f32(eta)
        self.thickness = max(f32(thickness), EPS)
        self.g = min(max(f32(g), f32(-0.99)), f32(0.99))
        self.max_depth, self.n_samples = int(max_depth), int(n_samples)


def _local(v, n, tg, bt):
    return V(dot(v, tg), dot(v, bt), dot(v, n))


def _frame(n):
    tg, bt = coordinate_system(np.asarray(n, np.float32)[None])
    return tg[0], bt[0]


def regularize_alpha(a):
    return min(max(f32(f32(2) * a), f32(0.1)), f32(0.3)) if a < f32(0.3) else a


def coated_sample(P, wo, n, u, uc, regularize=False):
    """sample_bsdf_spectral(::CoatedDiffuseMaterial) (:1232-1414) -> None, or (wi world, f [4], pdf, is_specular, eta)"""
    wo, n = np.asarray(wo, np.float32), np.asarray(n, np.float32)
    wdn = dot(wo, n)
    if abs(wdn) < f32(1e-6):
        return None
    ax, ay = (regularize_alpha(P.ax), regularize_alpha(P.ay)) if regularize else (P.ax, P.ay)
    tg, bt = _frame(n)
    wl = V(dot(wo, tg), dot(wo, bt), wdn)
    flip = wl[2] < 0
    if flip:
        wl = -wl
    th = P.thickness
    bs = sample_dielectric(wl, uc, u, ax, ay, P.eta, ALL)
    if (not bs.valid) or bs.pdf == 0 or bs.wi[2] == 0:
        return None

    def world(w):
        w = -w if flip else w
        return normalize((tg * w[0] + bt * w[1] + n * w[2]).astype(np.float32))
    if bs.refl:
        return world(bs.wi), bs.f, bs.pdf, bs.spec, f32(1)
    w, spec = bs.wi, bs.spec
    f = (bs.f * abs(w[2])).astype(np.float32)
    pdf = bs.pdf
    z = th
    rng = PCG32(hash_seed_vec(wl), hash_f_p2(uc, u))
    with np.errstate(divide="ignore", invalid="ignore"):
        for depth in range(P.max_depth):
            rr = f32(mx(f) / pdf)
            if depth > 3 and rr < f32(0.25):
                q = max(f32(0), f32(f32(1) - rr))
                if rng.f32() < q:
                    return None
                pdf = f32(pdf * f32(f32(1) - q))
            if w[2] == 0:
                return None
            if P.has_medium:
                dz = f32(-_r(np.log, f32(f32(1) - rng.f32())) / f32(f32(1) / abs(w[2])))
                zp = f32(z + dz) if w[2] > 0 else f32(z - dz)
                if zp == z:
                    return None
                if 0 < zp < th:
                    pu = (rng.f32(), rng.f32())
                    wi_p, pp = sample_hg(P.g, -w, pu)
                    if pp == 0 or wi_p[2] == 0:
                        return None
                    f = (f * P.albedo * pp).astype(np.float32)
                    pdf = f32(pdf * pp)
                    spec, w, z = False, wi_p, zp
                    continue
                z = min(max(zp, f32(0)), th)
            else:
                z = f32(0) if z == th else th
                f = (f * layer_tr(th, w)).astype(np.float32)
            at_bottom = z == 0
            uc2, u2 = rng.f32(), None
            u2 = (rng.f32(), rng.f32())
            bi = P.bottom.sample(-w, u2, uc2, ALL) if at_bottom else sample_dielectric(-w, uc2, u2, ax, ay, P.eta, ALL)
            if (not bi.valid) or bi.pdf == 0 or bi.wi[2] == 0:
                return None
            f = (f * bi.f).astype(np.float32)
            pdf = f32(pdf * bi.pdf)
            spec = spec and bi.spec
            w = bi.wi
            if not bi.refl:
                return world(w), f, pdf, spec, bi.eta
            f = (f * abs(bi.wi[2])).astype(np.float32)
    return None


def coated_pdf(P, wo, wi):
    """pdf_layered_bsdf (:1848-1928) on LOCAL directions (wo.z > 0 after the two-sided flip)"""
    ax, ay, eta = P.ax, P.ay, P.eta
    rng = PCG32(hash_seed_vec(wi), hash_vec(wo))
    same = same_hemisphere(wo, wi)
    smooth = tr_smooth(ax, ay)
    ns = f32(P.n_samples)
    total = f32(0)
    if same and not smooth:
        total = f32(total + f32(ns * pdf_dielectric(wo, wi, ax, ay, eta, REFL)))
    with np.errstate(divide="ignore", invalid="ignore"):
        for _ in range(P.n_samples):
            if same:
                uc1, u1 = rng.f32(), (rng.f32(), rng.f32())
                wos = sample_dielectric(wo, uc1, u1, ax, ay, eta, TRANS)
                uc2, u2 = rng.f32(), (rng.f32(), rng.f32())
                wis = sample_dielectric(wi, uc2, u2, ax, ay, eta, TRANS)
                if wos.valid and wos.pdf > 0 and wis.valid and wis.pdf > 0:
                    if smooth:
                        total = f32(total + P.bottom.pdf(-wos.wi, -wis.wi))
                    else:
                        u3, uc3 = P.bottom.draw_rs(rng)
                        rs = P.bottom.sample(-wos.wi, u3, uc3, ALL)
                        if rs.valid and rs.pdf > 0:
                            r_pdf = P.bottom.pdf(-wos.wi, -wis.wi)
                            total = f32(total + f32(power_heuristic(wis.pdf, r_pdf) * r_pdf))
                            t_pdf = pdf_dielectric(-rs.wi, wi, ax, ay, eta)
                            total = f32(total + f32(power_heuristic(rs.pdf, t_pdf) * t_pdf))
            else:
                uc1, u1 = rng.f32(), (rng.f32(), rng.f32())
                wos = sample_dielectric(wo, uc1, u1, ax, ay, eta, TRANS)
                if (not wos.valid) or wos.pdf == 0 or wos.refl:
                    continue
                u2, uc2 = P.bottom.draw_wis(rng)
                wis = P.bottom.sample(wi, u2, uc2, TRANS)
                if (not wis.valid) or wis.pdf == 0 or wis.refl:
                    continue
                if smooth:
                    total = f32(total + P.bottom.pdf(-wos.wi, wi))
                else:
                    total = f32(total + f32(f32(pdf_dielectric(wo, -wis.wi, ax, ay, eta) + P.bottom.pdf(-wos.wi, wi)) / f32(2)))
    est = f32(total / ns)
    # lerp(0.9f0, 1 / 4 pi, est) with the reference's OWN lerp(v1, v2, t) = (1 - t) v1 + t v2 (spectrum.jl:33): the estimate is the interpolation
    # PARAMETER between 0.9 and 1 / 4 pi — pbrt-v4 mixes the other way round (Lerp(0.9, 1 / 4 pi, estimate)); the reference is what is restated
    return f32(f32(f32(f32(1) - est) * f32(0.9)) + f32(est * f32(f32(1) / f32(f32(4) * PI))))


def coated_eval(P, wo, wi, n):
    """evaluate_bsdf_spectral(::CoatedDiffuseMaterial) (:1564-1836) -> (f [4], pdf)"""
    wo, wi, n = np.asarray(wo, np.float32), np.asarray(wi, np.float32), np.asarray(n, np.float32)
    ax, ay, eta, th, g = P.ax, P.ay, P.eta, P.thickness, P.g
    tg, bt = _frame(n)
    wol, wil = V(dot(wo, tg), dot(wo, bt), dot(wo, n)), V(dot(wi, tg), dot(wi, bt), dot(wi, n))
    if wol[2] < 0:
        wol, wil = -wol, -wil
    if abs(wol[2]) < f32(1e-6) or abs(wil[2]) < f32(1e-6):
        return np.zeros(4, np.float32), f32(0)
    same = same_hemisphere(wol, wil)
    exit_bottom = same != True      # same_hemi xor entered_top (entered_top = true)
    exit_z = f32(0) if exit_bottom else th
    res = np.zeros(4, np.float32)
    ns = f32(P.n_samples)
    if same:
        ef, _ = eval_dielectric(wol, wil, ax, ay, eta)
        res = (res + s4(ef) * ns).astype(np.float32)
    rng = PCG32(hash_seed_vec(wol), hash_vec(wil))
    smooth = tr_smooth(ax, ay)
    with np.errstate(divide="ignore", invalid="ignore"):
        for _ in range(P.n_samples):
            uc, u = rng.f32(), (rng.f32(), rng.f32())
            wos = sample_dielectric(wol, uc, u, ax, ay, eta, TRANS)
            if (not wos.valid) or wos.pdf == 0 or wos.wi[2] == 0:
                continue
            uc, u = rng.f32(), (rng.f32(), rng.f32())
            wis = P.bottom.sample(wil, u, uc, TRANS) if exit_bottom else sample_dielectric(wil, uc, u, ax, ay, eta, TRANS)
            if (not wis.valid) or wis.pdf == 0 or wis.wi[2] == 0:
                continue
            beta = (wos.f * abs(wos.wi[2]) / wos.pdf).astype(np.float32)
            z, w = th, wos.wi
            for depth in range(P.max_depth):
                if depth > 3 and mx(beta) < f32(0.25):
                    q = max(f32(0), f32(f32(1) - mx(beta)))
                    if rng.f32() < q:
                        break
                    beta = (beta / f32(f32(1) - q)).astype(np.float32)
                if P.has_medium:
                    dz = f32(-_r(np.log, f32(f32(1) - rng.f32())) / f32(f32(1) / abs(w[2])))
                    zp = f32(z + dz) if w[2] > 0 else f32(z - dz)
                    if zp == z:
                        continue
                    if 0 < zp < th:
                        cph = dot(-w, -wis.wi)
                        if exit_bottom:
                            wt = power_heuristic(wis.pdf, hg_pdf(g, cph))
                        else:
                            wt = power_heuristic(wis.pdf, hg_pdf(g, cph)) if not smooth else f32(1)
                        phase = hg_pdf(g, cph)
                        res = (res + beta * P.albedo * phase * wt * layer_tr(f32(zp - exit_z), wis.wi) * wis.f / wis.pdf).astype(np.float32)
                        pu = (rng.f32(), rng.f32())
                        wi_p, pp = sample_hg(g, -w, pu)
                        if pp == 0 or wi_p[2] == 0:
                            break
                        beta = (beta * P.albedo * pp / pp).astype(np.float32)
                        w, z = wi_p, zp
                        if (z < exit_z and w[2] > 0) or (z > exit_z and w[2] < 0):
                            if exit_bottom:
                                fe, epdf = P.bottom.eval(-w, wil)
                            else:
                                if not smooth:
                                    fv, _ = eval_dielectric(-w, wil, ax, ay, eta)
                                    fe, epdf = s4(fv), pdf_dielectric(-w, wil, ax, ay, eta, TRANS)
                                else:
                                    continue
                            if mx(fe) > 0:
                                res = (res + beta * layer_tr(f32(zp - exit_z), wi_p) * fe * power_heuristic(pp, epdf)).astype(np.float32)
                        continue
                    z = min(max(zp, f32(0)), th)
                else:
                    z = f32(0) if z == th else th
                    beta = (beta * layer_tr(th, w)).astype(np.float32)
                if z == exit_z:
                    uc, u = rng.f32(), (rng.f32(), rng.f32())
                    bs = P.bottom.sample(-w, u, uc, REFL) if exit_bottom else sample_dielectric(-w, uc, u, ax, ay, eta, REFL)
                    if (not bs.valid) or bs.pdf == 0 or bs.wi[2] == 0:
                        break
                    beta = (beta * bs.f * abs(bs.wi[2]) / bs.pdf).astype(np.float32)
                    w = bs.wi
                else:
                    non_exit_spec = smooth if z == th else False
                    if not non_exit_spec:
                        if z == th:
                            fv, _ = eval_dielectric(-w, -wis.wi, ax, ay, eta)
                            fn = s4(fv)
                        else:
                            fn, _ = P.bottom.eval(-w, -wis.wi)
                        if mx(fn) > 0:
                            wt = f32(1)
                            if (not exit_bottom) or (not smooth):
                                npdf = pdf_dielectric(-w, -wis.wi, ax, ay, eta) if z == th else P.bottom.pdf(-w, -wis.wi)
                                wt = power_heuristic(wis.pdf, npdf)
                            res = (res + beta * fn * abs(wis.wi[2]) * wt * layer_tr(th, wis.wi) * wis.f / wis.pdf).astype(np.float32)
                    uc, u = rng.f32(), (rng.f32(), rng.f32())
                    bs = sample_dielectric(-w, uc, u, ax, ay, eta, REFL) if z == th else P.bottom.sample(-w, u, uc, REFL)
                    if (not bs.valid) or bs.pdf == 0 or bs.wi[2] == 0:
                        break
                    beta = (beta * bs.f * abs(bs.wi[2]) / bs.pdf).astype(np.float32)
                    w = bs.wi
                    if (not smooth) or exit_bottom:
                        if exit_bottom:
                            f3, _ = P.bottom.eval(-w, wil)
                        else:
                            fv, _ = eval_dielectric(-w, wil, ax, ay, eta)
                            f3 = s4(fv)
                        if mx(f3) > 0:
                            wt3 = f32(1)
                            if not non_exit_spec:
                                e3 = P.bottom.pdf(-w, wil) if exit_bottom else pdf_dielectric(-w, wil, ax, ay, eta, TRANS)
                                wt3 = power_heuristic(bs.pdf, e3)
                            res = (res + beta * layer_tr(th, bs.wi) * f3 * wt3).astype(np.float32)
    res = (res / ns).astype(np.float32)
    return res, coated_pdf(P, wol, wil)


# ---------------------------------------------------------------------------------------------------- CoatedConductor (spectral-eval.jl:2877-3425)
# NOT a random walk in the reference: an analytic composition of the dielectric coating and the conductor base, case by case
# (smooth / rough coating x smooth / rough conductor).  Deterministic, no hashes.
class CoatedCond:
    """parameters at one wavelength set: interface eta (0 -> 1) and alphas, conductor alphas, conductor eta / k [4] ALREADY divided by the
    interface eta (:2933-2935), thickness (>= eps), albedo [4], has_medium"""

    def __init__(self, ieta, iax, iay, cax, cay, ce, ck, thickness, albedo, has_medium):
        self.ieta = f32(1) if f32(ieta) == 0 else f32(ieta)
        self.iax, self.iay, self.cax, self.cay = f32(iax), f32(iay), f32(cax), f32(cay)
        self.ce, self.ck = (np.asarray(ce, np.float32) / self.ieta).astype(np.float32), (np.asarray(ck, np.float32) / self.ieta).astype(np.float32)
        self.thickness = max(f32(thickness), EPS)
        self.albedo, self.has_medium = np.asarray(albedo, np.float32), bool(has_medium)


def _frc(c, P):
    from ref_volpath_np import fr_complex
    return fr_complex(np.array([[f32(c)]], np.float32), P.ce[None], P.ck[None])[0]


def cc_sample(P, wo, n, u, uc, regularize=False):
    """sample_bsdf_spectral(::CoatedConductorMaterial) -> None, or (wi world, f [4], pdf, is_specular)"""
    wo, n = np.asarray(wo, np.float32), np.asarray(n, np.float32)
    wdn = dot(wo, n)
    if abs(wdn) < f32(1e-6):
        return None
    iax, iay, cax, cay = P.iax, P.iay, P.cax, P.cay
    if regularize:
        iax, iay, cax, cay = regularize_alpha(iax), regularize_alpha(iay), regularize_alpha(cax), regularize_alpha(cay)
    ieta, th = P.ieta, P.thickness
    tg, bt = _frame(n)
    wl = V(dot(wo, tg), dot(wo, bt), wdn)
    flip = wl[2] < 0
    if flip:
        wl = -wl
    co = abs(wl[2])

    def world(w):
        w = -w if flip else w
        return normalize((tg * w[0] + bt * w[1] + n * w[2]).astype(np.float32))
    i_smooth, c_smooth = tr_smooth(iax, iay), tr_smooth(cax, cay)
    with np.errstate(divide="ignore", invalid="ignore"):
        if i_smooth:
            Fi = fresnel(co, ieta)
            if uc < Fi:
                return world(V(-wl[0], -wl[1], wl[2])), s4(1.0), f32(1), True
            s2t = f32(max(f32(0), f32(f32(1) - f32(co * co))) / f32(ieta * ieta))
            if s2t >= 1:
                return None
            ct_in = f32(np.sqrt(f32(f32(1) - s2t)))
            if c_smooth:
                wb = normalize(V(f32(-wl[0] / ieta), f32(-wl[1] / ieta), ct_in))
                Fc = _frc(ct_in, P)
                s2o = f32(max(f32(0), f32(f32(1) - f32(wb[2] * wb[2]))) * f32(ieta * ieta))
                if s2o >= 1:
                    return None
                c_out = f32(np.sqrt(f32(f32(1) - s2o)))
                T_in, T_out = f32(f32(1) - Fi), f32(f32(1) - fresnel(c_out, ieta))
                if P.has_medium:
                    tr = layer_tr(th, V(0, 0, ct_in))
                    ltr = (f32(tr * tr) * P.albedo).astype(np.float32)
                else:
                    ltr = s4(1.0)
                f = (Fc * T_in * T_out * ltr / co).astype(np.float32)
                return world(V(-wl[0], -wl[1], wl[2])), f, f32(f32(1) - Fi), True
            woc = normalize(V(f32(wl[0] / ieta), f32(wl[1] / ieta), ct_in))
            cax, cay = max(cax, f32(1e-4)), max(cay, f32(1e-4))
            wm = tr_sample_wm(woc, u, cax, cay)
            com = dot(woc, wm)
            if com < 0:
                return None
            wic = ((-woc).astype(np.float32) + (f32(f32(2) * com) * wm).astype(np.float32)).astype(np.float32)
            if wic[2] < 0:
                return None
            Fc = _frc(abs(com), P)
            fc = (f32(tr_d(wm, cax, cay)) * Fc * tr_g(woc, wic, cax, cay) / f32(f32(f32(4) * abs(woc[2])) * abs(wic[2]))).astype(np.float32)
            s2o = f32(f32(f32(wic[0] * wic[0]) + f32(wic[1] * wic[1])) * f32(ieta * ieta))
            if s2o >= 1:
                return None
            c_out = f32(np.sqrt(f32(f32(1) - s2o)))
            T_in, T_out = f32(f32(1) - Fi), f32(f32(1) - fresnel(c_out, ieta))
            if P.has_medium:
                ltr = (f32(layer_tr(th, V(0, 0, ct_in)) * layer_tr(th, V(0, 0, wic[2]))) * P.albedo).astype(np.float32)
            else:
                ltr = s4(1.0)
            wil = normalize(V(f32(wic[0] * ieta), f32(wic[1] * ieta), c_out))
            f = (fc * T_in * T_out * ltr).astype(np.float32)
            pdf = f32(f32(f32(1) - Fi) * f32(tr_pdf(woc, wm, cax, cay) / f32(f32(4) * abs(com))))
            return world(wil), f, pdf, False
        # rough coating
        iax, iay = max(iax, f32(1e-4)), max(iay, f32(1e-4))
        wm = tr_sample_wm(wl, u, iax, iay)
        com = dot(wl, wm)
        if com < 0:
            return None
        Fi = fresnel(com, ieta)
        if uc < Fi:
            wil = ((-wl).astype(np.float32) + (f32(f32(2) * com) * wm).astype(np.float32)).astype(np.float32)
            if f32(wil[2] * wl[2]) < 0:
                return None
            fv = f32(f32(tr_d(wm, iax, iay) * tr_g(wl, wil, iax, iay)) / f32(f32(f32(4) * abs(wil[2])) * abs(wl[2])))
            pdf = f32(f32(Fi * tr_pdf(wl, wm, iax, iay)) / f32(f32(4) * abs(com)))
            return world(wil), s4(fv), pdf, False
        T_in = f32(f32(1) - Fi)
        lc = V(-wl[0], -wl[1], wl[2])
        if c_smooth:
            cb = abs(lc[2])
            Fc = _frc(cb, P)
            T_out = f32(f32(1) - fresnel(cb, ieta))
            if P.has_medium:
                tr = layer_tr(th, lc)
                ltr = (f32(tr * tr) * P.albedo).astype(np.float32)
            else:
                ltr = s4(1.0)
            f = (Fc * T_in * T_out * ltr / co).astype(np.float32)
            pdf = f32(f32(f32(f32(1) - Fi) * tr_pdf(wl, wm, iax, iay)) / f32(f32(4) * abs(com)))
            return world(lc), f, pdf, False
        cax, cay = max(cax, f32(1e-4)), max(cay, f32(1e-4))
        wmc = tr_sample_wm(wl, u, cax, cay)
        comc = dot(wl, wmc)
        if comc < 0:
            return None
        wil = ((-wl).astype(np.float32) + (f32(f32(2) * comc) * wmc).astype(np.float32)).astype(np.float32)
        if f32(wil[2] * wl[2]) < 0:
            return None
        Fc = _frc(abs(comc), P)
        ci, c_o = abs(wil[2]), abs(wl[2])
        fc = (f32(tr_d(wmc, cax, cay)) * Fc * tr_g(wl, wil, cax, cay) / f32(f32(f32(4) * ci) * c_o)).astype(np.float32)
        T_out = f32(f32(1) - fresnel(ci, ieta))
        if P.has_medium:
            ltr = (f32(layer_tr(th, V(0, 0, c_o)) * layer_tr(th, wil)) * P.albedo).astype(np.float32)
        else:
            ltr = s4(1.0)
        f = (fc * T_in * T_out * ltr).astype(np.float32)
        pdf = f32(f32(f32(f32(1) - Fi) * tr_pdf(wl, wmc, cax, cay)) / f32(f32(4) * abs(comc)))
        return world(wil), f, pdf, False


def cc_eval(P, wo, wi, n):
    """evaluate_bsdf_spectral(::CoatedConductorMaterial) (:3243-3412) -> (f [4], pdf)"""
    wo, wi, n = np.asarray(wo, np.float32), np.asarray(wi, np.float32), np.asarray(n, np.float32)
    zero = (np.zeros(4, np.float32), f32(0))
    ci, co = dot(wi, n), dot(wo, n)
    if f32(ci * co) < 0:
        return zero
    if abs(ci) < f32(1e-6) or abs(co) < f32(1e-6):
        return zero
    iax, iay, cax, cay, ieta, th = P.iax, P.iay, P.cax, P.cay, P.ieta, P.thickness
    tg, bt = _frame(n)
    wol, wil = V(dot(wo, tg), dot(wo, bt), co), V(dot(wi, tg), dot(wi, bt), ci)
    if wol[2] < 0:
        wol, wil = -wol, -wil
    i_smooth, c_smooth = tr_smooth(iax, iay), tr_smooth(cax, cay)
    if i_smooth and c_smooth:
        return zero
    with np.errstate(divide="ignore", invalid="ignore"):
        wh = normalize((wol + wil).astype(np.float32))
        if wh[2] < 0:
            wh = -wh
        coh = dot(wol, wh)
        F_wh, F_o, F_i = fresnel(abs(coh), ieta), fresnel(abs(wol[2]), ieta), fresnel(abs(wil[2]), ieta)
        T_o, T_i = f32(f32(1) - F_o), f32(f32(1) - F_i)
        if P.has_medium:
            tr = layer_tr(th, wil)
            ltr = (f32(tr * tr) * P.albedo).astype(np.float32)
        else:
            ltr = s4(1.0)
        den = f32(f32(f32(4) * abs(wil[2])) * abs(wol[2]))
        if i_smooth:
            cax, cay = max(cax, f32(1e-4)), max(cay, f32(1e-4))
            fc = (f32(tr_d(wh, cax, cay)) * _frc(abs(coh), P) * tr_g(wol, wil, cax, cay) / den).astype(np.float32)
            f = (fc * T_o * T_i * ltr).astype(np.float32)
            pdf = f32(f32(T_o * tr_pdf(wol, wh, cax, cay)) / f32(f32(4) * abs(coh)))
            return f, pdf
        iax, iay = max(iax, f32(1e-4)), max(iay, f32(1e-4))
        f_int = f32(f32(f32(tr_d(wh, iax, iay) * F_wh) * tr_g(wol, wil, iax, iay)) / den)
        if c_smooth:
            fc = (_frc(abs(wol[2]), P) / abs(wol[2])).astype(np.float32)
            pdf_c = f32(1)
        else:
            cax, cay = max(cax, f32(1e-4)), max(cay, f32(1e-4))
            fc = (f32(tr_d(wh, cax, cay)) * _frc(abs(coh), P) * tr_g(wol, wil, cax, cay) / den).astype(np.float32)
            pdf_c = f32(tr_pdf(wol, wh, cax, cay) / f32(f32(4) * abs(coh)))
        f = (s4(f_int) + (fc * T_o * T_i * ltr).astype(np.float32)).astype(np.float32)
        pdf_i = f32(f32(F_o * tr_pdf(wol, wh, iax, iay)) / f32(f32(4) * abs(coh)))
        return f, f32(pdf_i + f32(T_o * pdf_c))

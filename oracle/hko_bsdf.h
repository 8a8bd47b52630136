// hko_bsdf.h — CPU ORACLE (test infrastructure): spectral BSDF sampling / evaluation.
// Follows src/materials/spectral-eval.jl:
//   Matte sample :42-101, eval :371-398 (sigma scales f in sample only: quirk Q12)
//   Mirror sample :108-132, eval :400-406
//   Glass sample :140-198 (smooth only), eval :408-414
//   Conductor sample :223-318, eval :422-488
//   generic fallback (gray 0.5 Lambertian, quirk Q24) :322-359, :491-511
//   helpers: coordinate_system :3514-3533, TR :3663-3754, fr_complex :3579-3647
//   fresnel_dielectric  src/reflection/bxdf.jl:67-100 ; roughness_to_alpha / regularize_alpha
//   src/reflection/microfacet.jl:83-99 ; PiecewiseLinearSpectrum  src/spectral/piecewise-linear.jl
//   MixMaterial resolve  src/materials/mix-material.jl:96-127, 146-163, 222-238
#pragma once
#include "hko_lights.h"
#include "hko_sampler.h"

namespace hko {

struct BSDFSample {
    V3 wi = V3(0, 0, 1);
    Spec f;
    float pdf = 0.0f;
    bool is_specular = false;
    float eta_scale = 1.0f;
};

struct MaterialCtx {
    const RGB2SpecTable* table;
    TextureSet textures;
    const hk_material* materials;
    int32_t n_materials;
    const hk_pl_spectrum* spectra;
};

inline void coordinate_system(V3 n, V3& tangent, V3& bitangent) {
    if (std::fabs(n.x) > std::fabs(n.y)) {
        float inv_len = 1.0f / std::sqrt(n.x * n.x + n.z * n.z);
        tangent = V3(n.z * inv_len, 0.0f, -n.x * inv_len);
    } else {
        float inv_len = 1.0f / std::sqrt(n.y * n.y + n.z * n.z);
        tangent = V3(0.0f, n.z * inv_len, -n.y * inv_len);
    }
    bitangent = cross(n, tangent);
}
inline V3 local_to_world(V3 l, V3 n, V3 t, V3 b) { return t * l.x + b * l.y + n * l.z; }
inline V3 world_to_local(V3 v, V3 n, V3 t, V3 b) { return V3(dot(v, t), dot(v, b), dot(v, n)); }
inline V3 reflect(V3 wo, V3 n) { return -wo + 2.0f * dot(wo, n) * n; }
inline bool same_hemisphere(V3 a, V3 b) { return a.z * b.z > 0.0f; }
inline float abs_cos_theta(V3 w) { return std::fabs(w.z); }
inline float cos2_theta(V3 w) { return w.z * w.z; }
inline float sin2_theta(V3 w) { return maxf(0.0f, 1.0f - cos2_theta(w)); }
inline float sin_theta(V3 w) { return std::sqrt(sin2_theta(w)); }
inline float tan2_theta(V3 w) { return sin2_theta(w) / cos2_theta(w); }
inline float cos_phi(V3 w) {
    float s = sin_theta(w);
    return s == 0.0f ? 1.0f : clampf(w.x / s, -1.0f, 1.0f);
}
inline float sin_phi(V3 w) {
    float s = sin_theta(w);
    return s == 0.0f ? 0.0f : clampf(w.y / s, -1.0f, 1.0f);
}
inline V3 face_forward(V3 v, V3 n) { return dot(v, n) < 0.0f ? -v : v; }

inline float fresnel_dielectric(float cos_i, float eta) {
    cos_i = clampf(cos_i, -1.0f, 1.0f);
    if (cos_i < 0.0f) {
        eta = 1.0f / eta;
        cos_i = -cos_i;
    }
    float sin2_i = 1.0f - cos_i * cos_i;
    float sin2_t = sin2_i / (eta * eta);
    if (sin2_t >= 1.0f) return 1.0f;
    float cos_t = std::sqrt(1.0f - sin2_t);
    float r_parl = (eta * cos_i - cos_t) / (eta * cos_i + cos_t);
    float r_perp = (cos_i - eta * cos_t) / (cos_i + eta * cos_t);
    return 0.5f * (r_parl * r_parl + r_perp * r_perp);
}
inline float roughness_to_alpha(float r) { return std::sqrt(r); }
inline float regularize_alpha(float a) { return a < 0.3f ? clampf(2.0f * a, 0.1f, 0.3f) : a; }
inline bool tr_smooth(float ax, float ay) { return maxf(ax, ay) < 1e-3f; }

inline float fr_complex(float cos_i, float eta, float k) {
    cos_i = clampf(cos_i, 0.0f, 1.0f);
    float sin2_i = 1.0f - cos_i * cos_i;
    float eta2 = eta * eta, k2 = k * k;
    float ec2_re = eta2 - k2, ec2_im = 2.0f * eta * k;
    float denom = ec2_re * ec2_re + ec2_im * ec2_im;
    float s2t_re = sin2_i * ec2_re / denom;
    float s2t_im = -sin2_i * ec2_im / denom;
    float c2t_re = 1.0f - s2t_re, c2t_im = -s2t_im;
    float mag = std::sqrt(c2t_re * c2t_re + c2t_im * c2t_im);
    float ct_re = std::sqrt(0.5f * (mag + c2t_re));
    float ct_im = c2t_im / (2.0f * ct_re);
    if (ct_re == 0.0f) ct_im = std::sqrt(0.5f * mag);
    float eci_re = eta * cos_i, eci_im = k * cos_i;
    float np_re = eci_re - ct_re, np_im = eci_im - ct_im;
    float dp_re = eci_re + ct_re, dp_im = eci_im + ct_im;
    float dp_m2 = dp_re * dp_re + dp_im * dp_im;
    float rp_re = (np_re * dp_re + np_im * dp_im) / dp_m2;
    float rp_im = (np_im * dp_re - np_re * dp_im) / dp_m2;
    float ect_re = eta * ct_re - k * ct_im, ect_im = eta * ct_im + k * ct_re;
    float ns_re = cos_i - ect_re, ns_im = -ect_im;
    float ds_re = cos_i + ect_re, ds_im = ect_im;
    float ds_m2 = ds_re * ds_re + ds_im * ds_im;
    float rs_re = (ns_re * ds_re + ns_im * ds_im) / ds_m2;
    float rs_im = (ns_im * ds_re - ns_re * ds_im) / ds_m2;
    float norm_parl = rp_re * rp_re + rp_im * rp_im;
    float norm_perp = rs_re * rs_re + rs_im * rs_im;
    return (norm_parl + norm_perp) * 0.5f;
}
inline Spec fr_complex_spectral(float c, const Spec& eta, const Spec& k) {
    return Spec(fr_complex(c, eta[0], k[0]), fr_complex(c, eta[1], k[1]), fr_complex(c, eta[2], k[2]), fr_complex(c, eta[3], k[3]));
}
inline float tr_d(V3 wm, float ax, float ay) {
    float t2 = tan2_theta(wm);
    if (std::isinf(t2)) return 0.0f;
    float c4 = cos2_theta(wm) * cos2_theta(wm);
    if (c4 < 1e-16f) return 0.0f;
    float a = cos_phi(wm) / ax, b = sin_phi(wm) / ay;
    float e = t2 * (a * a + b * b);
    return 1.0f / (PI_F * ax * ay * c4 * ((1.0f + e) * (1.0f + e)));
}
inline float tr_lambda(V3 w, float ax, float ay) {
    float t2 = tan2_theta(w);
    if (std::isinf(t2)) return 0.0f;
    float a = cos_phi(w) * ax, b = sin_phi(w) * ay;
    float alpha2 = a * a + b * b;
    return (std::sqrt(1.0f + alpha2 * t2) - 1.0f) * 0.5f;
}
inline float tr_g1(V3 w, float ax, float ay) { return 1.0f / (1.0f + tr_lambda(w, ax, ay)); }
inline float tr_g(V3 wo, V3 wi, float ax, float ay) { return 1.0f / (1.0f + tr_lambda(wo, ax, ay) + tr_lambda(wi, ax, ay)); }
inline float tr_pdf(V3 w, V3 wm, float ax, float ay) { return tr_g1(w, ax, ay) / abs_cos_theta(w) * tr_d(wm, ax, ay) * std::fabs(dot(w, wm)); }
inline V3 tr_sample_wm(V3 w, V2 u, float ax, float ay) {
    V3 wh = normalize(V3(ax * w.x, ay * w.y, w.z));
    if (wh.z < 0.0f) wh = -wh;
    V3 t1 = wh.z < 0.99999f ? normalize(cross(V3(0, 0, 1), wh)) : V3(1, 0, 0);
    V3 t2 = cross(wh, t1);
    float r = std::sqrt(u.x);
    float phi = 2.0f * PI_F * u.y;
    float sphi, cphi;
    jl_sincos(phi, sphi, cphi);
    float px = r * cphi, py = r * sphi;
    float h = std::sqrt(1.0f - px * px);
    py = lerpf(h, py, 0.5f * (1.0f + wh.z));
    float pz = std::sqrt(maxf(0.0f, 1.0f - px * px - py * py));
    V3 nh = px * t1 + py * t2 + pz * wh;
    return normalize(V3(ax * nh.x, ay * nh.y, maxf(1e-6f, nh.z)));
}

inline float pl_sample(const hk_pl_spectrum& s, float lam) {
    int N = s.n;
    if (lam <= s.lambdas[0]) return s.values[0];
    if (lam >= s.lambdas[N - 1]) return s.values[N - 1];
    int lo = 1, hi = N;
    while (lo + 1 < hi) {
        int mid = (lo + hi) >> 1;
        if (s.lambdas[mid - 1] <= lam)
            lo = mid;
        else
            hi = mid;
    }
    float t = (lam - s.lambdas[lo - 1]) / (s.lambdas[hi - 1] - s.lambdas[lo - 1]);
    return s.values[lo - 1] * (1.0f - t) + s.values[hi - 1] * t;
}
// eval_ior_spectral  spectral-eval.jl:204-208
inline Spec eval_ior(const MaterialCtx& c, const hk_material& m, int slot, const TexCtx& uv, const Wavelengths& w) {
    if (m.spectrum[slot] >= 0) {
        const hk_pl_spectrum& s = c.spectra[m.spectrum[slot]];
        return Spec(pl_sample(s, w.lambda[0]), pl_sample(s, w.lambda[1]), pl_sample(s, w.lambda[2]), pl_sample(s, w.lambda[3]));
    }
    return uplift_rgb_unbounded(*c.table, eval_tex(c.textures, m.rgb[slot], uv), w);
}

inline BSDFSample sample_lambert(V3 wo, V3 n, V2 u, const Spec& f_scaled) {
    BSDFSample s;
    float wo_dot_n = dot(wo, n);
    V3 tangent, bitangent;
    coordinate_system(n, tangent, bitangent);
    V3 lw = cosine_sample_hemisphere(u);
    float cos_theta = lw.z;
    if (cos_theta < 1e-6f) return s;
    if (wo_dot_n < 0.0f) lw = V3(lw.x, lw.y, -lw.z);
    V3 wi = normalize(local_to_world(lw, n, tangent, bitangent));
    s.wi = wi;
    s.f = f_scaled;
    s.pdf = cos_theta / PI_F;
    s.is_specular = false;
    s.eta_scale = 1.0f;
    return s;
}

// sample_bsdf_spectral dispatch (material-dispatch.jl:23-31)
inline BSDFSample sample_bsdf(const MaterialCtx& c, int32_t mat_idx, V3 wo_world, V3 n, const TexCtx& uv, const Wavelengths& w, V2 u, float rng, bool regularize) {
    const hk_material& m = c.materials[mat_idx];
    const RGB2SpecTable& T = *c.table;
    switch (m.kind) {
        case HK_MAT_MATTE: {
            float wo_dot_n = dot(wo_world, n);
            if (std::fabs(wo_dot_n) < 1e-6f) return BSDFSample();
            RGBA kd = clamp_rgb(eval_tex(c.textures, m.rgb[0], uv));
            float sigma = eval_tex(c.textures, m.f[0], uv);
            Spec kds = uplift_rgb(T, kd, w);
            Spec f;
            if (sigma > 0.0f) {
                float rf = 1.0f - 0.5f * sigma / (sigma + 0.33f);
                f = kds * (rf / PI_F);
            } else
                f = kds * (1.0f / PI_F);
            return sample_lambert(wo_world, n, u, f);
        }
        case HK_MAT_MIRROR: {
            float wo_dot_n = dot(wo_world, n);
            if (std::fabs(wo_dot_n) < 1e-6f) return BSDFSample();
            Spec kr = uplift_rgb(T, eval_tex(c.textures, m.rgb[0], uv), w);
            V3 no = wo_dot_n < 0.0f ? -n : n;
            BSDFSample s;
            s.wi = reflect(wo_world, no);
            s.f = kr;
            s.pdf = 1.0f;
            s.is_specular = true;
            return s;
        }
        case HK_MAT_GLASS: {
            RGBA kr_rgb = eval_tex(c.textures, m.rgb[0], uv), kt_rgb = eval_tex(c.textures, m.rgb[1], uv);
            float ior = eval_tex(c.textures, m.f[0], uv);
            if (ior == 0.0f) ior = 1.0f;
            Spec kr = uplift_rgb(T, kr_rgb, w), kt = uplift_rgb(T, kt_rgb, w);
            float cos_o = dot(wo_world, n);
            bool entering = cos_o > 0.0f;
            V3 no = entering ? n : -n;
            cos_o = std::fabs(cos_o);
            float eta = entering ? ior : (1.0f / ior);
            float F = fresnel_dielectric(cos_o, eta);
            BSDFSample s;
            s.pdf = 1.0f;
            s.is_specular = true;
            if (rng < F) {
                s.wi = reflect(wo_world, no);
                s.f = kr;
                return s;
            }
            float sin2_i = maxf(0.0f, 1.0f - cos_o * cos_o);
            float sin2_t = sin2_i / (eta * eta);
            if (sin2_t >= 1.0f) {
                s.wi = reflect(wo_world, no);
                s.f = kr;
                return s;
            }
            float cos_t = std::sqrt(1.0f - sin2_t);
            s.wi = normalize(-wo_world / eta + (cos_o / eta - cos_t) * no);
            s.f = kt;
            s.eta_scale = 1.0f / (eta * eta);
            return s;
        }
        case HK_MAT_CONDUCTOR: {
            V3 tangent, bitangent;
            coordinate_system(n, tangent, bitangent);
            V3 wo = world_to_local(wo_world, n, tangent, bitangent);
            if (wo.z == 0.0f) return BSDFSample();
            float roughness = eval_tex(c.textures, m.f[0], uv);
            float ax = (m.flags & HK_MATF_REMAP_ROUGHNESS) ? roughness_to_alpha(roughness) : roughness;
            float ay = ax;
            if (regularize) {
                ax = regularize_alpha(ax);
                ay = regularize_alpha(ay);
            }
            if (!tr_smooth(ax, ay)) {
                ax = maxf(ax, 1e-4f);
                ay = maxf(ay, 1e-4f);
            }
            Spec eta = eval_ior(c, m, 0, uv, w), k = eval_ior(c, m, 1, uv, w);
            BSDFSample s;
            if (tr_smooth(ax, ay)) {
                V3 wi(-wo.x, -wo.y, wo.z);
                float ci = abs_cos_theta(wi);
                Spec F = fr_complex_spectral(ci, eta, k);
                s.f = F / ci;
                s.wi = local_to_world(wi, n, tangent, bitangent);
                s.pdf = 1.0f;
                s.is_specular = true;
                return s;
            }
            V3 wm = tr_sample_wm(wo, u, ax, ay);
            V3 wi = -wo + 2.0f * dot(wo, wm) * wm;
            if (!same_hemisphere(wo, wi)) return BSDFSample();
            float pdf = tr_pdf(wo, wm, ax, ay) / (4.0f * std::fabs(dot(wo, wm)));
            float co = abs_cos_theta(wo), ci = abs_cos_theta(wi);
            if (ci == 0.0f || co == 0.0f) return BSDFSample();
            Spec F = fr_complex_spectral(std::fabs(dot(wo, wm)), eta, k);
            float D = tr_d(wm, ax, ay), G = tr_g(wo, wi, ax, ay);
            s.f = D * F * G / (4.0f * ci * co);
            s.wi = local_to_world(wi, n, tangent, bitangent);
            s.pdf = pdf;
            s.is_specular = false;
            return s;
        }
        default: {  // generic fallback: gray 0.5 Lambertian
            float wo_dot_n = dot(wo_world, n);
            if (std::fabs(wo_dot_n) < 1e-6f) return BSDFSample();
            return sample_lambert(wo_world, n, u, Spec(0.5f) * (1.0f / PI_F));
        }
    }
}

// evaluate_bsdf_spectral dispatch (material-dispatch.jl:46-53): returns f, pdf
inline Spec eval_bsdf(const MaterialCtx& c, int32_t mat_idx, V3 wo_world, V3 wi_world, V3 n, const TexCtx& uv, const Wavelengths& w, float& pdf) {
    const hk_material& m = c.materials[mat_idx];
    const RGB2SpecTable& T = *c.table;
    pdf = 0.0f;
    switch (m.kind) {
        case HK_MAT_MATTE: {
            float ci = dot(wi_world, n), co = dot(wo_world, n);
            if (ci * co < 0.0f) return Spec();
            float ct = std::fabs(ci);
            if (ct < 1e-6f) return Spec();
            RGBA kd = clamp_rgb(eval_tex(c.textures, m.rgb[0], uv));
            Spec kds = uplift_rgb(T, kd, w);
            pdf = ct / PI_F;
            return kds / PI_F;
        }
        case HK_MAT_MIRROR:
        case HK_MAT_GLASS: return Spec();
        case HK_MAT_CONDUCTOR: {
            V3 tangent, bitangent;
            coordinate_system(n, tangent, bitangent);
            V3 wo = world_to_local(wo_world, n, tangent, bitangent);
            V3 wi = world_to_local(wi_world, n, tangent, bitangent);
            if (!same_hemisphere(wo, wi)) return Spec();
            float roughness = eval_tex(c.textures, m.f[0], uv);
            float ax = (m.flags & HK_MATF_REMAP_ROUGHNESS) ? roughness_to_alpha(roughness) : roughness;
            float ay = ax;
            if (!tr_smooth(ax, ay)) {
                ax = maxf(ax, 1e-4f);
                ay = maxf(ay, 1e-4f);
            }
            if (tr_smooth(ax, ay)) return Spec();
            float co = abs_cos_theta(wo), ci = abs_cos_theta(wi);
            if (ci == 0.0f || co == 0.0f) return Spec();
            V3 wm = wi + wo;
            if (dot(wm, wm) == 0.0f) return Spec();
            wm = normalize(wm);
            Spec eta = eval_ior(c, m, 0, uv, w), k = eval_ior(c, m, 1, uv, w);
            Spec F = fr_complex_spectral(std::fabs(dot(wo, wm)), eta, k);
            float D = tr_d(wm, ax, ay), G = tr_g(wo, wi, ax, ay);
            Spec f = D * F * G / (4.0f * ci * co);
            V3 wmp = face_forward(wm, V3(0, 0, 1));
            pdf = tr_pdf(wo, wmp, ax, ay) / (4.0f * std::fabs(dot(wo, wmp)));
            return f;
        }
        default: {
            float ci = dot(wi_world, n), co = dot(wo_world, n);
            if (ci * co < 0.0f) return Spec();
            float ct = std::fabs(ci);
            if (ct < 1e-6f) return Spec();
            pdf = ct / PI_F;
            return Spec(0.5f / PI_F);
        }
    }
}

// get_surface_alpha (spectral-eval.jl:3882-3888): Matte -> Kd alpha, everything else 1
inline float surface_alpha(const MaterialCtx& c, int32_t mat_idx, V2 uv) {
    const hk_material& m = c.materials[mat_idx];
    if (m.kind == HK_MAT_MATTE) return eval_tex_nearest(c.textures, m.rgb[0], uv).c[3];  // Point2f method => nearest texel (quirk Q28)
    return 1.0f;
}

// mix_hash_float  mix-material.jl:96-127
inline float mix_hash_float(V3 p, V3 wo, const uint32_t key[4]) {
    uint64_t h = 0;
    h ^= (uint64_t)f2u(p.x);
    h *= 0xcc9e2d51ull;
    h ^= (uint64_t)(uint32_t)(f2u(p.y) << 4);
    h *= 0x1b873593ull;
    h ^= (uint64_t)(uint32_t)(f2u(p.z) << 8);
    h ^= (uint64_t)(uint32_t)(f2u(wo.x) << 16);
    h *= 0xcc9e2d51ull;
    h ^= (uint64_t)f2u(wo.y);
    h *= 0x1b873593ull;
    h ^= (uint64_t)(uint32_t)(f2u(wo.z) << 12);
    h ^= (uint64_t)key[0] << 24;
    h ^= (uint64_t)key[1];
    h *= 0xcc9e2d51ull;
    h ^= (uint64_t)key[2] << 28;
    h ^= (uint64_t)key[3] << 4;
    h *= 0x1b873593ull;
    h ^= h >> 31;
    h *= 0x7fb5d329728ea185ull;
    h ^= h >> 27;
    h *= 0x81dadef4bc2dd44dull;
    h ^= h >> 33;
    return (float)(uint32_t)(h & 0xFFFFFFFFull) * 2.3283064365386963e-10f;
}
// resolve_mix_material  mix-material.jl:222-238
inline int32_t resolve_mix_material(const MaterialCtx& c, int32_t idx, V3 p, V3 wo, V2 uv) {
    int32_t cur = idx;
    for (int it = 0; it < 8; ++it) {
        const hk_material& m = c.materials[cur];
        if (m.kind != HK_MAT_MIX) return cur;
        float amt = eval_tex_nearest(c.textures, m.f[0], uv);  // eval_tex(ctx, mix.amount, uv::Point2f): nearest texel (Q28)
        if (amt <= 0.0f)
            cur = m.i[0];
        else if (amt >= 1.0f)
            cur = m.i[1];
        else {
            float u = mix_hash_float(p, wo, m.mix_key);
            cur = amt < u ? m.i[0] : m.i[1];
        }
    }
    return cur;
}

}  // namespace hko

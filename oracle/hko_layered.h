// hko_layered.h — CPU ORACLE (test infrastructure): layered / two-sided-transmissive materials.
// Follows src/materials/spectral-eval.jl:
//   layer_transmittance :836-839, sample_hg_phase_spectral :846-873, hg_phase_pdf :880-884,
//   sample_dielectric_interface :965-1046, refract_pbrt :1055-1077, refract_microfacet :1084-1104,
//   sample/eval/pdf_diffuse_interface :1143-1198, power_heuristic :1205-1214,
//   eval_dielectric_interface :1449-1503, pdf_dielectric_interface :1510-1575,
//   CoatedDiffuse sample :1233-1441, eval :1563-1832, pdf_layered_bsdf :1840-1937 ("simplified" PDF, kept as is)
//   ThinDielectric :1975-2051 ; DiffuseTransmission :2083-2218
//   diffuse-transmission bottom :2253-2337 ; CoatedDiffuseTransmission sample :2341-2497, eval :2501-2744,
//   pdf_layered_bsdf_dt :2748-2832 ; CoatedConductor sample :2877-3231 (analytic 2-lobe), eval :3238-3420
// The random walks are seeded from float bit patterns (pbrt_hash(seed, wo_local) ...), so a walk is
// reproducible only when wo_local / wi_local are bit-identical.
#pragma once
#include "hko_bsdf.h"

namespace hko {

enum { BX_R = 1, BX_T = 2, BX_ALL = 3 };

struct LSample {
    Spec f;
    V3 wi = V3(0.0f);
    float pdf = 0.0f;
    bool is_reflection = false, is_specular = false;
    float eta = 1.0f;
    bool valid = false;
};
inline LSample lsample(const Spec& f, V3 wi, float pdf, bool refl, bool spec, float eta) {
    LSample s;
    s.f = f, s.wi = wi, s.pdf = pdf, s.is_reflection = refl, s.is_specular = spec, s.eta = eta, s.valid = true;
    return s;
}

inline float layer_tr(float thickness, V3 w) {
    if (std::fabs(thickness) <= 1.1920929e-7f) return 1.0f;
    return std::exp(-std::fabs(thickness / w.z));
}
inline float hg_phase_pdf(float g, float cos_t) {
    float g2 = g * g;
    float denom = 1.0f + g2 - 2.0f * g * cos_t;
    return (1.0f - g2) / (4.0f * PI_F * denom * std::sqrt(maxf(1e-10f, denom)));
}
inline V3 sample_hg_phase(float g, V3 wo, V2 u, float& p) {
    float cos_t;
    if (std::fabs(g) < 1e-3f)
        cos_t = 1.0f - 2.0f * u.x;
    else {
        float g2 = g * g;
        float sq = (1.0f - g2) / (1.0f - g + 2.0f * g * u.x);
        cos_t = clampf((1.0f + g2 - sq * sq) / (2.0f * g), -1.0f, 1.0f);
    }
    float sin_t = std::sqrt(maxf(0.0f, 1.0f - cos_t * cos_t));
    float phi = 2.0f * PI_F * u.y;
    V3 t1, t2;
    coordinate_system(-wo, t1, t2);
    float sphi, cphi;
    jl_sincos(phi, sphi, cphi);
    V3 wi = sin_t * cphi * t1 + sin_t * sphi * t2 + cos_t * (-wo);
    wi = normalize(wi);
    p = hg_phase_pdf(g, cos_t);
    return wi;
}
inline float sample_exponential(float u, float a) { return -std::log(1.0f - u) / a; }
inline float power_heuristic1(float fp, float gp) {
    float f2 = fp * fp, g2 = gp * gp;
    if (f2 + g2 == 0.0f) return 0.0f;
    return f2 / (f2 + g2);
}

inline bool refract_pbrt(V3 wo, float eta, V3& wi, float& etap) {
    float ci = wo.z;
    etap = ci > 0.0f ? eta : (1.0f / eta);
    float s2i = maxf(0.0f, 1.0f - ci * ci);
    float s2t = s2i / (etap * etap);
    if (s2t >= 1.0f) return false;
    float ct = std::sqrt(1.0f - s2t);
    float cts = ci > 0.0f ? -ct : ct;
    wi = normalize(V3(-wo.x / etap, -wo.y / etap, cts));
    return true;
}
inline bool refract_microfacet(V3 wo, V3 wm, float eta, V3& wi, float& etap) {
    float ci = dot(wo, wm);
    etap = ci > 0.0f ? eta : (1.0f / eta);
    float s2i = maxf(0.0f, 1.0f - ci * ci);
    float s2t = s2i / (etap * etap);
    if (s2t >= 1.0f) return false;
    float ct = std::sqrt(1.0f - s2t);
    float cts = ci > 0.0f ? -ct : ct;
    wi = normalize(-wo / etap + (ci / etap + cts) * wm);
    return true;
}

// ---- top interface: DielectricBxDF --------------------------------------------------------------
inline LSample sample_dielectric_interface(V3 wo, float uc, V2 u, float ax, float ay, float eta, int flags) {
    if (tr_smooth(ax, ay) || eta == 1.0f) {
        float R = fresnel_dielectric(wo.z, eta), T = 1.0f - R;
        float pr = (flags & BX_R) ? R : 0.0f, pt = (flags & BX_T) ? T : 0.0f;
        if (pr == 0.0f && pt == 0.0f) return LSample();
        if (uc < pr / (pr + pt)) {
            V3 wi(-wo.x, -wo.y, wo.z);
            return lsample(Spec(R / std::fabs(wi.z)), wi, pr / (pr + pt), true, true, 1.0f);
        }
        V3 wi;
        float etap;
        if (!refract_pbrt(wo, eta, wi, etap)) return LSample();
        return lsample(Spec(T / std::fabs(wi.z)), wi, pt / (pr + pt), false, true, etap);
    }
    V3 wm = tr_sample_wm(wo, u, ax, ay);
    float com = dot(wo, wm);
    float R = fresnel_dielectric(com, eta), T = 1.0f - R;
    float pr = (flags & BX_R) ? R : 0.0f, pt = (flags & BX_T) ? T : 0.0f;
    if (pr == 0.0f && pt == 0.0f) return LSample();
    if (uc < pr / (pr + pt)) {
        V3 wi = reflect(wo, wm);
        if (!same_hemisphere(wo, wi)) return LSample();
        float pdf = tr_pdf(wo, wm, ax, ay) / (4.0f * std::fabs(com)) * pr / (pr + pt);
        float D = tr_d(wm, ax, ay), G = tr_g(wo, wi, ax, ay);
        float f = D * G * R / (4.0f * wo.z * wi.z);
        return lsample(Spec(f), wi, pdf, true, false, 1.0f);
    }
    V3 wi;
    float etap;
    if (!refract_microfacet(wo, wm, eta, wi, etap) || same_hemisphere(wo, wi) || wi.z == 0.0f) return LSample();
    float s = dot(wi, wm) + dot(wo, wm) / etap;
    float denom = s * s;
    float dwm_dwi = std::fabs(dot(wi, wm)) / denom;
    float pdf = tr_pdf(wo, wm, ax, ay) * dwm_dwi * pt / (pr + pt);
    float D = tr_d(wm, ax, ay), G = tr_g(wo, wi, ax, ay);
    float f = T * D * G * std::fabs(dot(wi, wm) * dot(wo, wm) / (wi.z * wo.z * denom));
    return lsample(Spec(f), wi, pdf, false, false, etap);
}
// only f is consumed by the callers (the reference discards the pdf of this function)
inline Spec eval_dielectric_interface(V3 wo, V3 wi, float ax, float ay, float eta) {
    if (tr_smooth(ax, ay) || eta == 1.0f) return Spec();
    if (same_hemisphere(wo, wi)) {
        V3 wh = normalize(wo + wi);
        if (wh.z < 0.0f) wh = -wh;
        float R = fresnel_dielectric(dot(wo, wh), eta);
        float D = tr_d(wh, ax, ay), G = tr_g(wo, wi, ax, ay);
        return Spec(D * G * R / (4.0f * wo.z * wi.z));
    }
    float etap = wo.z > 0.0f ? eta : (1.0f / eta);
    V3 wh = normalize(wo + wi * etap);
    if (wh.z < 0.0f) wh = -wh;
    float coh = dot(wo, wh), cih = dot(wi, wh);
    if (coh * cih > 0.0f) return Spec();
    float R = fresnel_dielectric(coh, eta), T = 1.0f - R;
    float s = cih + coh / etap;
    float denom = s * s;
    float D = tr_d(wh, ax, ay), G = tr_g(wo, wi, ax, ay);
    return Spec(T * D * G * std::fabs(cih * coh / (wo.z * wi.z * denom)));
}
inline float pdf_dielectric_interface(V3 wo, V3 wi, float ax, float ay, float eta, int flags = BX_ALL) {
    if (tr_smooth(ax, ay) || eta == 1.0f) return 0.0f;
    if (same_hemisphere(wo, wi)) {
        if (!(flags & BX_R)) return 0.0f;
        V3 wh = normalize(wo + wi);
        if (wh.z < 0.0f) wh = -wh;
        float coh = std::fabs(dot(wo, wh));
        float R = fresnel_dielectric(coh, eta), T = 1.0f - R;
        float pr = (flags & BX_R) ? R : 0.0f, pt = (flags & BX_T) ? T : 0.0f;
        float pdf = tr_pdf(wo, wh, ax, ay) / (4.0f * coh);
        return pdf * pr / (pr + pt);
    }
    if (!(flags & BX_T)) return 0.0f;
    float etap = wo.z > 0.0f ? eta : (1.0f / eta);
    V3 wh = normalize(wo + wi * etap);
    if (wh.z < 0.0f) wh = -wh;
    float coh = dot(wo, wh), cih = dot(wi, wh);
    if (coh * cih > 0.0f) return 0.0f;
    float R = fresnel_dielectric(std::fabs(coh), eta), T = 1.0f - R;
    float pr = (flags & BX_R) ? R : 0.0f, pt = (flags & BX_T) ? T : 0.0f;
    float s = cih + coh / etap;
    float denom = s * s;
    float dwm_dwi = std::fabs(cih) / denom;
    float pdf = tr_pdf(wo, wh, ax, ay) * dwm_dwi;
    return pdf * pt / (pr + pt);
}

// ---- bottom interface: DiffuseBxDF (CoatedDiffuse) or DiffuseTransmissionBxDF ---------------------
struct Bottom {
    bool dt = false;  // false: sample/eval/pdf_diffuse_interface ; true: *_diffuse_transmission_bottom
    Spec refl, trans;
    float pr_max = 0.0f, pt_max = 0.0f;
};
inline LSample bottom_sample(const Bottom& b, V3 wo, V2 u, float uc, int flags) {
    if (!b.dt) {
        if (!(flags & BX_R)) return LSample();
        V3 wi = cosine_sample_hemisphere(u);
        if (wo.z < 0.0f) wi = V3(wi.x, wi.y, -wi.z);
        float ci = std::fabs(wi.z);
        if (ci < 1e-6f) return LSample();
        return lsample(b.refl * (1.0f / PI_F), wi, ci / PI_F, true, false, 1.0f);
    }
    float pr = (flags & BX_R) ? b.pr_max : 0.0f, pt = (flags & BX_T) ? b.pt_max : 0.0f;
    if (pr + pt < 1e-10f) return LSample();
    float prob_r = pr / (pr + pt);
    V3 wi = cosine_sample_hemisphere(u);
    if (uc < prob_r) {
        if (wo.z < 0.0f) wi = V3(wi.x, wi.y, -wi.z);
        float ci = std::fabs(wi.z);
        if (ci < 1e-6f) return LSample();
        return lsample(b.refl * (1.0f / PI_F), wi, prob_r * ci / PI_F, true, false, 1.0f);
    }
    if (wo.z > 0.0f) wi = V3(wi.x, wi.y, -wi.z);
    float ci = std::fabs(wi.z);
    if (ci < 1e-6f) return LSample();
    return lsample(b.trans * (1.0f / PI_F), wi, (1.0f - prob_r) * ci / PI_F, false, false, 1.0f);
}
inline Spec bottom_eval(const Bottom& b, V3 wo, V3 wi, float& pdf) {
    pdf = 0.0f;
    if (!b.dt) {
        if (!same_hemisphere(wo, wi)) return Spec();
        pdf = std::fabs(wi.z) / PI_F;
        return b.refl * (1.0f / PI_F);
    }
    if (b.pr_max + b.pt_max < 1e-10f) return Spec();
    float aci = std::fabs(wi.z);
    if (same_hemisphere(wo, wi)) {
        pdf = (b.pr_max / (b.pr_max + b.pt_max)) * aci / PI_F;
        return b.refl * (1.0f / PI_F);
    }
    pdf = (b.pt_max / (b.pr_max + b.pt_max)) * aci / PI_F;
    return b.trans * (1.0f / PI_F);
}
inline float bottom_pdf(const Bottom& b, V3 wo, V3 wi, int flags = BX_ALL) {
    if (!b.dt) {
        if (!same_hemisphere(wo, wi)) return 0.0f;
        return std::fabs(wi.z) / PI_F;
    }
    float pr = (flags & BX_R) ? b.pr_max : 0.0f, pt = (flags & BX_T) ? b.pt_max : 0.0f;
    if (pr + pt < 1e-10f) return 0.0f;
    float aci = std::fabs(wi.z);
    if (same_hemisphere(wo, wi)) return (pr / (pr + pt)) * aci / PI_F;
    return (pt / (pr + pt)) * aci / PI_F;
}

struct LayeredParams {
    Bottom bottom;
    float ax, ay, eta, thickness, g;
    Spec albedo;
    bool has_medium;
    int max_depth, n_samples;
};

// LayeredBxDF::Sample_f as the reference restates it (CoatedDiffuse :1233-1441, CoatedDiffuseTransmission :2341-2497)
inline BSDFSample layered_sample(const LayeredParams& P, V3 wo_world, V3 n, V2 sample_u, float rng_in) {
    float wo_dot_n = dot(wo_world, n);
    if (std::fabs(wo_dot_n) < 1e-6f) return BSDFSample();
    V3 tangent, bitangent;
    coordinate_system(n, tangent, bitangent);
    V3 wo = V3(dot(wo_world, tangent), dot(wo_world, bitangent), wo_dot_n);
    bool flip = wo.z < 0.0f;
    if (flip) wo = -wo;
    LSample bs = sample_dielectric_interface(wo, rng_in, sample_u, P.ax, P.ay, P.eta, BX_ALL);
    if (!bs.valid || bs.pdf == 0.0f || bs.wi.z == 0.0f) return BSDFSample();
    BSDFSample out;
    if (bs.is_reflection) {
        V3 wl = flip ? -bs.wi : bs.wi;
        out.wi = normalize(tangent * wl.x + bitangent * wl.y + n * wl.z);
        out.f = bs.f, out.pdf = bs.pdf, out.is_specular = bs.is_specular, out.eta_scale = 1.0f;
        return out;
    }
    V3 w = bs.wi;
    bool specular_path = bs.is_specular;
    Spec f = bs.f * std::fabs(w.z);
    float pdf = bs.pdf;
    float z = P.thickness;
    PCG32 rng = pcg32_init(pbrt_hash((uint64_t)0, wo), pbrt_hash(rng_in, sample_u));
    for (int depth = 0; depth < P.max_depth; ++depth) {
        float rr_beta = max_component(f) / pdf;
        if (depth > 3 && rr_beta < 0.25f) {
            float q = maxf(0.0f, 1.0f - rr_beta);
            if (pcg32_uniform_f32(rng) < q) return BSDFSample();
            pdf *= 1.0f - q;
        }
        if (w.z == 0.0f) return BSDFSample();
        if (P.has_medium) {
            float dz = sample_exponential(pcg32_uniform_f32(rng), 1.0f / std::fabs(w.z));
            float zp = w.z > 0.0f ? (z + dz) : (z - dz);
            if (zp == z) return BSDFSample();
            if (0.0f < zp && zp < P.thickness) {
                float u1 = pcg32_uniform_f32(rng), u2 = pcg32_uniform_f32(rng);
                float pp;
                V3 wp = sample_hg_phase(P.g, -w, V2(u1, u2), pp);
                if (pp == 0.0f || wp.z == 0.0f) return BSDFSample();
                f = f * P.albedo * pp;
                pdf *= pp;
                specular_path = false;
                w = wp;
                z = zp;
                continue;
            }
            z = clampf(zp, 0.0f, P.thickness);
        } else {
            z = (z == P.thickness) ? 0.0f : P.thickness;
            f = f * layer_tr(P.thickness, w);
        }
        float uc = pcg32_uniform_f32(rng), u1 = pcg32_uniform_f32(rng), u2 = pcg32_uniform_f32(rng);
        LSample bi = (z == 0.0f) ? bottom_sample(P.bottom, -w, V2(u1, u2), uc, BX_ALL)
                                 : sample_dielectric_interface(-w, uc, V2(u1, u2), P.ax, P.ay, P.eta, BX_ALL);
        if (!bi.valid || bi.pdf == 0.0f || bi.wi.z == 0.0f) return BSDFSample();
        f = f * bi.f;
        pdf *= bi.pdf;
        specular_path = specular_path && bi.is_specular;
        w = bi.wi;
        if (!bi.is_reflection) {
            V3 wl = flip ? -w : w;
            out.wi = normalize(tangent * wl.x + bitangent * wl.y + n * wl.z);
            out.f = f, out.pdf = pdf, out.is_specular = specular_path, out.eta_scale = bi.eta;
            return out;
        }
        f = f * std::fabs(bi.wi.z);
    }
    return BSDFSample();
}

// pdf_layered_bsdf :1840-1937 / pdf_layered_bsdf_dt :2748-2832
inline float layered_pdf(const LayeredParams& P, V3 wo, V3 wi) {
    PCG32 rng = pcg32_init(pbrt_hash((uint64_t)0, wi), pbrt_hash(wo));
    bool same_hemi = same_hemisphere(wo, wi);
    bool is_smooth = tr_smooth(P.ax, P.ay);
    const Bottom& B = P.bottom;
    float pdf_sum = 0.0f;
    if (same_hemi) {
        if (!is_smooth)
            pdf_sum += (float)P.n_samples * pdf_dielectric_interface(wo, wi, P.ax, P.ay, P.eta, BX_R);
    }
    for (int s = 0; s < P.n_samples; ++s) {
        if (same_hemi) {
            float uc1 = pcg32_uniform_f32(rng), u1 = pcg32_uniform_f32(rng), u2 = pcg32_uniform_f32(rng);
            LSample wos = sample_dielectric_interface(wo, uc1, V2(u1, u2), P.ax, P.ay, P.eta, BX_T);
            float uc2 = pcg32_uniform_f32(rng), u3 = pcg32_uniform_f32(rng), u4 = pcg32_uniform_f32(rng);
            LSample wis = sample_dielectric_interface(wi, uc2, V2(u3, u4), P.ax, P.ay, P.eta, BX_T);
            if (wos.valid && wos.pdf > 0.0f && wis.valid && wis.pdf > 0.0f) {
                if (is_smooth)
                    pdf_sum += bottom_pdf(B, -wos.wi, -wis.wi);
                else {
                    float u5 = pcg32_uniform_f32(rng), u6 = pcg32_uniform_f32(rng);
                    float uc3 = B.dt ? pcg32_uniform_f32(rng) : 0.0f;
                    LSample rs = bottom_sample(B, -wos.wi, V2(u5, u6), uc3, BX_ALL);
                    if (rs.valid && rs.pdf > 0.0f) {
                        float r_pdf = bottom_pdf(B, -wos.wi, -wis.wi);
                        pdf_sum += power_heuristic1(wis.pdf, r_pdf) * r_pdf;
                        float t_pdf = pdf_dielectric_interface(-rs.wi, wi, P.ax, P.ay, P.eta);
                        pdf_sum += power_heuristic1(rs.pdf, t_pdf) * t_pdf;
                    }
                }
            }
        } else {
            float uc1 = pcg32_uniform_f32(rng), u1 = pcg32_uniform_f32(rng), u2 = pcg32_uniform_f32(rng);
            LSample wos = sample_dielectric_interface(wo, uc1, V2(u1, u2), P.ax, P.ay, P.eta, BX_T);
            if (!wos.valid || wos.pdf == 0.0f || wos.is_reflection) continue;
            float uc2 = B.dt ? pcg32_uniform_f32(rng) : 0.0f;
            float u3 = pcg32_uniform_f32(rng), u4 = pcg32_uniform_f32(rng);
            LSample wis = bottom_sample(B, wi, V2(u3, u4), uc2, BX_T);
            if (!wis.valid || wis.pdf == 0.0f || wis.is_reflection) continue;
            if (is_smooth)
                pdf_sum += bottom_pdf(B, -wos.wi, wi);
            else
                pdf_sum += (pdf_dielectric_interface(wo, -wis.wi, P.ax, P.ay, P.eta) + bottom_pdf(B, -wos.wi, wi)) / 2.0f;
        }
    }
    // quirk Q26: Hikari's lerp is (v1, v2, t) (spectrum.jl:33) but is called with pbrt's (t, a, b) argument order,
    // so the returned density is (1 - p)*0.9 + p/(4 pi) with p = pdf_sum/n_samples.  Kept as the reference computes it.
    return lerpf(0.9f, 1.0f / (4.0f * PI_F), pdf_sum / (float)P.n_samples);
}

// LayeredBxDF::f as the reference restates it (CoatedDiffuse :1563-1832, CoatedDiffuseTransmission :2501-2744)
inline Spec layered_eval(const LayeredParams& P, V3 wo_world, V3 wi_world, V3 n, float& pdf) {
    pdf = 0.0f;
    const Bottom& B = P.bottom;
    V3 tangent, bitangent;
    coordinate_system(n, tangent, bitangent);
    float co = dot(wo_world, n), ci = dot(wi_world, n);
    V3 wo(dot(wo_world, tangent), dot(wo_world, bitangent), co);
    V3 wi(dot(wi_world, tangent), dot(wi_world, bitangent), ci);
    if (wo.z < 0.0f) {
        wo = -wo;
        wi = -wi;
    }
    if (std::fabs(wo.z) < 1e-6f || std::fabs(wi.z) < 1e-6f) return Spec();
    bool same_hemi = same_hemisphere(wo, wi);
    bool exit_at_bottom = !same_hemi;  // same_hemi XOR entered_top(=true)
    float exit_z = exit_at_bottom ? 0.0f : P.thickness;
    Spec fr;
    if (same_hemi) fr = fr + eval_dielectric_interface(wo, wi, P.ax, P.ay, P.eta) * (float)P.n_samples;
    PCG32 rng = pcg32_init(pbrt_hash((uint64_t)0, wo), pbrt_hash(wi));
    bool is_smooth = tr_smooth(P.ax, P.ay);
    for (int s = 0; s < P.n_samples; ++s) {
        float uc = pcg32_uniform_f32(rng), u1 = pcg32_uniform_f32(rng), u2 = pcg32_uniform_f32(rng);
        LSample wos = sample_dielectric_interface(wo, uc, V2(u1, u2), P.ax, P.ay, P.eta, BX_T);
        if (!wos.valid || wos.pdf == 0.0f || wos.wi.z == 0.0f) continue;
        uc = pcg32_uniform_f32(rng), u1 = pcg32_uniform_f32(rng), u2 = pcg32_uniform_f32(rng);
        LSample wis = exit_at_bottom ? bottom_sample(B, wi, V2(u1, u2), uc, BX_T)
                                     : sample_dielectric_interface(wi, uc, V2(u1, u2), P.ax, P.ay, P.eta, BX_T);
        if (!wis.valid || wis.pdf == 0.0f || wis.wi.z == 0.0f) continue;
        Spec beta = wos.f * std::fabs(wos.wi.z) / wos.pdf;
        float z = P.thickness;
        V3 w = wos.wi;
        for (int depth = 0; depth < P.max_depth; ++depth) {
            if (depth > 3 && max_component(beta) < 0.25f) {
                float q = maxf(0.0f, 1.0f - max_component(beta));
                if (pcg32_uniform_f32(rng) < q) break;
                beta = beta / (1.0f - q);
            }
            if (P.has_medium) {
                float dz = sample_exponential(pcg32_uniform_f32(rng), 1.0f / std::fabs(w.z));
                float zp = w.z > 0.0f ? (z + dz) : (z - dz);
                if (zp == z) continue;
                if (0.0f < zp && zp < P.thickness) {
                    float wt = 1.0f;
                    if (exit_at_bottom || !is_smooth) wt = power_heuristic1(wis.pdf, hg_phase_pdf(P.g, dot(-w, -wis.wi)));
                    float phase_val = hg_phase_pdf(P.g, dot(-w, -wis.wi));
                    fr = fr + beta * P.albedo * phase_val * wt * layer_tr(zp - exit_z, wis.wi) * wis.f / wis.pdf;
                    float pu1 = pcg32_uniform_f32(rng), pu2 = pcg32_uniform_f32(rng);
                    float pp;
                    V3 wp = sample_hg_phase(P.g, -w, V2(pu1, pu2), pp);
                    if (pp == 0.0f || wp.z == 0.0f) break;
                    beta = beta * P.albedo * pp / pp;
                    w = wp;
                    z = zp;
                    if ((z < exit_z && w.z > 0.0f) || (z > exit_z && w.z < 0.0f)) {
                        Spec fe;
                        float epdf;
                        if (exit_at_bottom)
                            fe = bottom_eval(B, -w, wi, epdf);
                        else {
                            if (is_smooth) continue;
                            fe = eval_dielectric_interface(-w, wi, P.ax, P.ay, P.eta);
                            epdf = pdf_dielectric_interface(-w, wi, P.ax, P.ay, P.eta, BX_T);
                        }
                        if (max_component(fe) > 0.0f) {
                            float wt2 = power_heuristic1(pp, epdf);
                            fr = fr + beta * layer_tr(zp - exit_z, wp) * fe * wt2;
                        }
                    }
                    continue;
                }
                z = clampf(zp, 0.0f, P.thickness);
            } else {
                z = (z == P.thickness) ? 0.0f : P.thickness;
                beta = beta * layer_tr(P.thickness, w);
            }
            if (z == exit_z) {
                uc = pcg32_uniform_f32(rng), u1 = pcg32_uniform_f32(rng), u2 = pcg32_uniform_f32(rng);
                LSample b2 = exit_at_bottom ? bottom_sample(B, -w, V2(u1, u2), uc, BX_R)
                                            : sample_dielectric_interface(-w, uc, V2(u1, u2), P.ax, P.ay, P.eta, BX_R);
                if (!b2.valid || b2.pdf == 0.0f || b2.wi.z == 0.0f) break;
                beta = beta * b2.f * std::fabs(b2.wi.z) / b2.pdf;
                w = b2.wi;
            } else {
                bool non_exit_is_bottom = (z == 0.0f);
                bool non_exit_is_specular = !non_exit_is_bottom && is_smooth;
                if (!non_exit_is_specular) {
                    float dummy;
                    Spec f_nee = non_exit_is_bottom ? bottom_eval(B, -w, -wis.wi, dummy) : eval_dielectric_interface(-w, -wis.wi, P.ax, P.ay, P.eta);
                    if (max_component(f_nee) > 0.0f) {
                        float wt = 1.0f;
                        if (!exit_at_bottom || !is_smooth) {
                            float nee_pdf = non_exit_is_bottom ? bottom_pdf(B, -w, -wis.wi) : pdf_dielectric_interface(-w, -wis.wi, P.ax, P.ay, P.eta);
                            wt = power_heuristic1(wis.pdf, nee_pdf);
                        }
                        fr = fr + beta * f_nee * std::fabs(wis.wi.z) * wt * layer_tr(P.thickness, wis.wi) * wis.f / wis.pdf;
                    }
                }
                uc = pcg32_uniform_f32(rng), u1 = pcg32_uniform_f32(rng), u2 = pcg32_uniform_f32(rng);
                LSample b2 = non_exit_is_bottom ? bottom_sample(B, -w, V2(u1, u2), uc, BX_R)
                                                : sample_dielectric_interface(-w, uc, V2(u1, u2), P.ax, P.ay, P.eta, BX_R);
                if (!b2.valid || b2.pdf == 0.0f || b2.wi.z == 0.0f) break;
                beta = beta * b2.f * std::fabs(b2.wi.z) / b2.pdf;
                w = b2.wi;
                if (!is_smooth || exit_at_bottom) {
                    float dummy;
                    Spec fe = exit_at_bottom ? bottom_eval(B, -w, wi, dummy) : eval_dielectric_interface(-w, wi, P.ax, P.ay, P.eta);
                    if (max_component(fe) > 0.0f) {
                        float wt3 = 1.0f;
                        if (!non_exit_is_specular) {
                            float epdf = exit_at_bottom ? bottom_pdf(B, -w, wi) : pdf_dielectric_interface(-w, wi, P.ax, P.ay, P.eta, BX_T);
                            wt3 = power_heuristic1(b2.pdf, epdf);
                        }
                        fr = fr + beta * layer_tr(P.thickness, b2.wi) * fe * wt3;
                    }
                }
            }
        }
    }
    fr = fr / (float)P.n_samples;
    pdf = layered_pdf(P, wo, wi);
    return fr;
}

inline float remap_alpha(const hk_material& m, float r) { return (m.flags & HK_MATF_REMAP_ROUGHNESS) ? roughness_to_alpha(r) : r; }
inline RGBA clamp01_rgb(const RGBA& a) { return RGBA(clampf(a.c[0], 0.0f, 1.0f), clampf(a.c[1], 0.0f, 1.0f), clampf(a.c[2], 0.0f, 1.0f), a.c[3]); }
inline bool rgb_is_black(const RGBA& a) { return a.c[0] == 0.0f && a.c[1] == 0.0f && a.c[2] == 0.0f; }
inline float max3(const RGBA& a) { return maxf(maxf(a.c[0], a.c[1]), a.c[2]); }

// parameter gathering for HK_MAT_COATED_DIFFUSE / HK_MAT_COATED_DIFFUSE_TRANSMISSION
inline LayeredParams layered_params(const MaterialCtx& c, const hk_material& m, const TexCtx& uv, const Wavelengths& w, bool regularize) {
    const RGB2SpecTable& T = *c.table;
    LayeredParams P;
    bool dt = m.kind == HK_MAT_COATED_DIFFUSE_TRANSMISSION;
    RGBA refl = eval_tex(c.textures, m.rgb[0], uv);
    RGBA albedo = eval_tex(c.textures, m.rgb[dt ? 2 : 1], uv);
    P.eta = eval_tex(c.textures, m.f[3], uv);
    P.thickness = maxf(eval_tex(c.textures, m.f[2], uv), 1.1920929e-7f);
    P.g = clampf(eval_tex(c.textures, m.f[4], uv), -0.99f, 0.99f);
    P.ax = remap_alpha(m, eval_tex(c.textures, m.f[0], uv));
    P.ay = remap_alpha(m, eval_tex(c.textures, m.f[1], uv));
    if (regularize) {
        P.ax = regularize_alpha(P.ax);
        P.ay = regularize_alpha(P.ay);
    }
    P.bottom.dt = dt;
    if (dt) {
        RGBA trans = eval_tex(c.textures, m.rgb[1], uv);
        refl = clamp01_rgb(refl);
        trans = clamp01_rgb(trans);
        P.bottom.trans = uplift_rgb(T, trans, w);
        P.bottom.pr_max = max3(refl);
        P.bottom.pt_max = max3(trans);
    }
    P.bottom.refl = uplift_rgb(T, refl, w);
    P.albedo = uplift_rgb(T, albedo, w);
    P.has_medium = !rgb_is_black(albedo);
    P.max_depth = m.i[0];
    P.n_samples = m.i[1];
    return P;
}

// ---- CoatedConductor (analytic two-lobe form of the reference, :2877-3420) -------------------------
struct CCParams {
    float ieta, iax, iay, cax, cay, thickness;
    Spec ce, ck, albedo;
    bool has_medium;
};
inline CCParams cc_params(const MaterialCtx& c, const hk_material& m, const TexCtx& uv, const Wavelengths& w, bool regularize) {
    const RGB2SpecTable& T = *c.table;
    CCParams P;
    P.ieta = eval_tex(c.textures, m.f[2], uv);
    if (P.ieta == 0.0f) P.ieta = 1.0f;
    P.iax = remap_alpha(m, eval_tex(c.textures, m.f[0], uv));
    P.iay = remap_alpha(m, eval_tex(c.textures, m.f[1], uv));
    P.cax = remap_alpha(m, eval_tex(c.textures, m.f[3], uv));
    P.cay = remap_alpha(m, eval_tex(c.textures, m.f[4], uv));
    if (regularize) {
        P.iax = regularize_alpha(P.iax), P.iay = regularize_alpha(P.iay);
        P.cax = regularize_alpha(P.cax), P.cay = regularize_alpha(P.cay);
    }
    if (m.flags & HK_MATF_USE_ETA_K) {
        P.ce = eval_ior(c, m, 0, uv, w);
        P.ck = eval_ior(c, m, 1, uv, w);
    } else {
        RGBA r = eval_tex(c.textures, m.rgb[2], uv);
        r = RGBA(clampf(r.c[0], 0.0f, 0.9999f), clampf(r.c[1], 0.0f, 0.9999f), clampf(r.c[2], 0.0f, 0.9999f), r.c[3]);
        Spec rs = uplift_rgb(T, r, w);
        P.ce = Spec(1.0f);
        Spec om = Spec(1.0f) - rs;
        for (int i = 0; i < 4; ++i) om.v[i] = maxf(om.v[i], 0.0f);  // clamp_zero
        P.ck = (2.0f * sqrt(rs)) / sqrt(om + Spec(1e-6f));
    }
    P.ce = P.ce / P.ieta;
    P.ck = P.ck / P.ieta;
    P.thickness = maxf(eval_tex(c.textures, m.f[5], uv), 1.1920929e-7f);
    RGBA albedo = eval_tex(c.textures, m.rgb[3], uv);
    P.albedo = uplift_rgb(T, albedo, w);
    P.has_medium = !rgb_is_black(albedo);
    return P;
}
inline BSDFSample cc_sample(CCParams P, V3 wo_world, V3 n, V2 sample_u, float rng) {
    float wo_dot_n = dot(wo_world, n);
    if (std::fabs(wo_dot_n) < 1e-6f) return BSDFSample();
    V3 tangent, bitangent;
    coordinate_system(n, tangent, bitangent);
    V3 wo(dot(wo_world, tangent), dot(wo_world, bitangent), wo_dot_n);
    bool flip = wo.z < 0.0f;
    if (flip) wo = -wo;
    float cos_o = std::fabs(wo.z);
    bool i_smooth = tr_smooth(P.iax, P.iay), c_smooth = tr_smooth(P.cax, P.cay);
    BSDFSample out;
    out.eta_scale = 1.0f;
    auto to_world = [&](V3 wl) { return normalize(tangent * wl.x + bitangent * wl.y + n * wl.z); };
    if (i_smooth) {
        float Fi = fresnel_dielectric(cos_o, P.ieta);
        if (rng < Fi) {
            V3 wl(-wo.x, -wo.y, wo.z);
            if (flip) wl = -wl;
            out.wi = to_world(wl), out.f = Spec(1.0f), out.pdf = 1.0f, out.is_specular = true;
            return out;
        }
        float s2t = maxf(0.0f, 1.0f - cos_o * cos_o) / (P.ieta * P.ieta);
        if (s2t >= 1.0f) return BSDFSample();
        float ct_in = std::sqrt(1.0f - s2t);
        if (c_smooth) {
            V3 wb = normalize(V3(-wo.x / P.ieta, -wo.y / P.ieta, ct_in));
            Spec Fc = fr_complex_spectral(ct_in, P.ce, P.ck);
            float s2o = maxf(0.0f, 1.0f - wb.z * wb.z) * (P.ieta * P.ieta);
            if (s2o >= 1.0f) return BSDFSample();
            float c_out = std::sqrt(1.0f - s2o);
            float Fo = fresnel_dielectric(c_out, P.ieta);
            float T_in = 1.0f - Fi, T_out = 1.0f - Fo;
            Spec ltr(1.0f);
            if (P.has_medium) {
                float tr = layer_tr(P.thickness, V3(0, 0, ct_in));
                ltr = tr * tr * P.albedo;
            }
            V3 wl(-wo.x, -wo.y, wo.z);
            if (flip) wl = -wl;
            out.wi = to_world(wl);
            out.f = Fc * T_in * T_out * ltr / cos_o;
            out.pdf = 1.0f - Fi, out.is_specular = true;
            return out;
        }
        V3 woc = normalize(V3(wo.x / P.ieta, wo.y / P.ieta, ct_in));
        float cax = maxf(P.cax, 1e-4f), cay = maxf(P.cay, 1e-4f);
        V3 wm = tr_sample_wm(woc, sample_u, cax, cay);
        float com = dot(woc, wm);
        if (com < 0.0f) return BSDFSample();
        V3 wic = -woc + 2.0f * com * wm;
        if (wic.z < 0.0f) return BSDFSample();
        Spec Fc = fr_complex_spectral(std::fabs(com), P.ce, P.ck);
        float D = tr_d(wm, cax, cay), G = tr_g(woc, wic, cax, cay);
        Spec fc = D * Fc * G / (4.0f * std::fabs(woc.z) * std::fabs(wic.z));
        float s2o = (wic.x * wic.x + wic.y * wic.y) * (P.ieta * P.ieta);
        if (s2o >= 1.0f) return BSDFSample();
        float c_out = std::sqrt(1.0f - s2o);
        float Fo = fresnel_dielectric(c_out, P.ieta);
        float T_in = 1.0f - Fi, T_out = 1.0f - Fo;
        Spec ltr(1.0f);
        if (P.has_medium) {
            float tr_in = layer_tr(P.thickness, V3(0, 0, ct_in)), tr_out = layer_tr(P.thickness, V3(0, 0, wic.z));
            ltr = tr_in * tr_out * P.albedo;
        }
        V3 wl = normalize(V3(wic.x * P.ieta, wic.y * P.ieta, c_out));
        if (flip) wl = -wl;
        out.wi = to_world(wl);
        out.f = fc * T_in * T_out * ltr;
        float pdf_m = tr_pdf(woc, wm, cax, cay);
        out.pdf = (1.0f - Fi) * (pdf_m / (4.0f * std::fabs(com)));
        out.is_specular = false;
        return out;
    }
    float iax = maxf(P.iax, 1e-4f), iay = maxf(P.iay, 1e-4f);
    V3 wm = tr_sample_wm(wo, sample_u, iax, iay);
    float com = dot(wo, wm);
    if (com < 0.0f) return BSDFSample();
    float Fi = fresnel_dielectric(com, P.ieta);
    if (rng < Fi) {
        V3 wl = -wo + 2.0f * com * wm;
        if (wl.z * wo.z < 0.0f) return BSDFSample();
        V3 wlf = flip ? -wl : wl;
        out.wi = to_world(wlf);
        // the reference evaluates D/G/cos on the (possibly flipped) wi_local: |cos| and TR terms are even in w
        float D = tr_d(wm, iax, iay), G = tr_g(wo, wlf, iax, iay);
        float ci = std::fabs(wlf.z), co = std::fabs(wo.z);
        float pdf_m = tr_pdf(wo, wm, iax, iay);
        out.pdf = Fi * pdf_m / (4.0f * std::fabs(com));
        out.f = Spec(D * G / (4.0f * ci * co));
        out.is_specular = false;
        return out;
    }
    float T_in = 1.0f - Fi;
    V3 lc(-wo.x, -wo.y, wo.z);
    if (c_smooth) {
        float cb = std::fabs(lc.z);
        Spec Fc = fr_complex_spectral(cb, P.ce, P.ck);
        float T_out = 1.0f - fresnel_dielectric(cb, P.ieta);
        Spec ltr(1.0f);
        if (P.has_medium) {
            float tr = layer_tr(P.thickness, lc);
            ltr = tr * tr * P.albedo;
        }
        if (flip) lc = -lc;
        out.wi = to_world(lc);
        out.f = Fc * T_in * T_out * ltr / cos_o;
        float pdf_m = tr_pdf(wo, wm, iax, iay);
        out.pdf = (1.0f - Fi) * pdf_m / (4.0f * std::fabs(com));
        out.is_specular = false;
        return out;
    }
    float cax = maxf(P.cax, 1e-4f), cay = maxf(P.cay, 1e-4f);
    V3 wmc = tr_sample_wm(wo, sample_u, cax, cay);
    float comc = dot(wo, wmc);
    if (comc < 0.0f) return BSDFSample();
    V3 wl = -wo + 2.0f * comc * wmc;
    if (wl.z * wo.z < 0.0f) return BSDFSample();
    Spec Fc = fr_complex_spectral(std::fabs(comc), P.ce, P.ck);
    float D = tr_d(wmc, cax, cay), G = tr_g(wo, wl, cax, cay);
    float ci = std::fabs(wl.z), co = std::fabs(wo.z);
    Spec fc = D * Fc * G / (4.0f * ci * co);
    float T_out = 1.0f - fresnel_dielectric(ci, P.ieta);
    Spec ltr(1.0f);
    if (P.has_medium) {
        float tr_in = layer_tr(P.thickness, V3(0, 0, co)), tr_out = layer_tr(P.thickness, wl);
        ltr = tr_in * tr_out * P.albedo;
    }
    if (flip) wl = -wl;
    out.wi = to_world(wl);
    out.f = fc * T_in * T_out * ltr;
    float pdf_m = tr_pdf(wo, wmc, cax, cay);
    out.pdf = (1.0f - Fi) * pdf_m / (4.0f * std::fabs(comc));
    out.is_specular = false;
    return out;
}
inline Spec cc_eval(CCParams P, V3 wo_world, V3 wi_world, V3 n, float& pdf) {
    pdf = 0.0f;
    float ci = dot(wi_world, n), co = dot(wo_world, n);
    if (ci * co < 0.0f) return Spec();
    if (std::fabs(ci) < 1e-6f || std::fabs(co) < 1e-6f) return Spec();
    V3 tangent, bitangent;
    coordinate_system(n, tangent, bitangent);
    V3 wo(dot(wo_world, tangent), dot(wo_world, bitangent), co);
    V3 wi(dot(wi_world, tangent), dot(wi_world, bitangent), ci);
    if (wo.z < 0.0f) {
        wo = -wo;
        wi = -wi;
    }
    bool i_smooth = tr_smooth(P.iax, P.iay), c_smooth = tr_smooth(P.cax, P.cay);
    if (i_smooth && c_smooth) return Spec();
    V3 wh = normalize(wo + wi);
    if (wh.z < 0.0f) wh = -wh;
    float coh = dot(wo, wh);
    float F_wh = fresnel_dielectric(std::fabs(coh), P.ieta);
    float F_o = fresnel_dielectric(std::fabs(wo.z), P.ieta);
    float F_i = fresnel_dielectric(std::fabs(wi.z), P.ieta);
    float T_o = 1.0f - F_o, T_i = 1.0f - F_i;
    Spec ltr(1.0f);
    if (P.has_medium) {
        float tr = layer_tr(P.thickness, wi);
        ltr = tr * tr * P.albedo;
    }
    if (i_smooth) {
        float cax = maxf(P.cax, 1e-4f), cay = maxf(P.cay, 1e-4f);
        float D = tr_d(wh, cax, cay), G = tr_g(wo, wi, cax, cay);
        Spec Fc = fr_complex_spectral(std::fabs(coh), P.ce, P.ck);
        Spec fc = D * Fc * G / (4.0f * std::fabs(wi.z) * std::fabs(wo.z));
        float pdf_m = tr_pdf(wo, wh, cax, cay);
        pdf = T_o * pdf_m / (4.0f * std::fabs(coh));
        return fc * T_o * T_i * ltr;
    }
    float iax = maxf(P.iax, 1e-4f), iay = maxf(P.iay, 1e-4f);
    float D_i = tr_d(wh, iax, iay), G_i = tr_g(wo, wi, iax, iay);
    float f_interface = D_i * F_wh * G_i / (4.0f * std::fabs(wi.z) * std::fabs(wo.z));
    Spec fc;
    float pdf_c;
    if (c_smooth) {
        Spec Fc = fr_complex_spectral(std::fabs(wo.z), P.ce, P.ck);
        fc = Fc / std::fabs(wo.z);
        pdf_c = 1.0f;
    } else {
        float cax = maxf(P.cax, 1e-4f), cay = maxf(P.cay, 1e-4f);
        float D_c = tr_d(wh, cax, cay), G_c = tr_g(wo, wi, cax, cay);
        Spec Fc = fr_complex_spectral(std::fabs(coh), P.ce, P.ck);
        fc = D_c * Fc * G_c / (4.0f * std::fabs(wi.z) * std::fabs(wo.z));
        pdf_c = tr_pdf(wo, wh, cax, cay) / (4.0f * std::fabs(coh));
    }
    Spec contrib = fc * T_o * T_i * ltr;
    float pdf_i = F_o * tr_pdf(wo, wh, iax, iay) / (4.0f * std::fabs(coh));
    pdf = pdf_i + T_o * pdf_c;
    return Spec(f_interface) + contrib;
}

// ---- ThinDielectric :1975-2051 ---------------------------------------------------------------------
inline BSDFSample thin_dielectric_sample(float eta, V3 wo_world, V3 n, float rng) {
    float wo_dot_n = dot(wo_world, n);
    if (std::fabs(wo_dot_n) < 1e-6f) return BSDFSample();
    V3 tangent, bitangent;
    coordinate_system(n, tangent, bitangent);
    V3 wo(dot(wo_world, tangent), dot(wo_world, bitangent), wo_dot_n);
    float cos_o = std::fabs(wo.z);
    float R0 = fresnel_dielectric(cos_o, eta), T0 = 1.0f - R0;
    float R = R0;
    if (R0 < 1.0f) R = R0 + T0 * T0 * R0 / (1.0f - R0 * R0);
    float T = 1.0f - R;
    if (R + T < 1e-10f) return BSDFSample();
    float prob_r = R / (R + T);
    BSDFSample s;
    s.is_specular = true, s.eta_scale = 1.0f;
    if (rng < prob_r) {
        V3 wl(-wo.x, -wo.y, wo.z);
        s.wi = normalize(tangent * wl.x + bitangent * wl.y + n * wl.z);
        s.f = Spec(R / std::fabs(wl.z));
        s.pdf = prob_r;
        return s;
    }
    s.wi = -wo_world;
    s.f = Spec(T / cos_o);
    s.pdf = 1.0f - prob_r;
    return s;
}

// ---- DiffuseTransmission :2083-2218 ----------------------------------------------------------------
struct DTParams {
    Spec r, t;
    float pr, pt;
};
inline DTParams dt_params(const MaterialCtx& c, const hk_material& m, const TexCtx& uv, const Wavelengths& w) {
    float scale = eval_tex(c.textures, m.f[0], uv);
    RGBA r = clamp01_rgb(eval_tex(c.textures, m.rgb[0], uv) * scale), t = clamp01_rgb(eval_tex(c.textures, m.rgb[1], uv) * scale);
    DTParams P;
    P.r = uplift_rgb(*c.table, r, w);
    P.t = uplift_rgb(*c.table, t, w);
    P.pr = max3(r);
    P.pt = max3(t);
    return P;
}
inline BSDFSample dt_sample(const DTParams& P, V3 wo_world, V3 n, V2 u, float rng) {
    float wo_dot_n = dot(wo_world, n);
    if (std::fabs(wo_dot_n) < 1e-6f) return BSDFSample();
    if (P.pr + P.pt < 1e-10f) return BSDFSample();
    V3 tangent, bitangent;
    coordinate_system(n, tangent, bitangent);
    float prob_r = P.pr / (P.pr + P.pt);
    bool refl = rng < prob_r;
    V3 lw = cosine_sample_hemisphere(u);
    if (refl ? (wo_dot_n < 0.0f) : (wo_dot_n > 0.0f)) lw = V3(lw.x, lw.y, -lw.z);
    float ct = std::fabs(lw.z);
    if (ct < 1e-6f) return BSDFSample();
    BSDFSample s;
    s.wi = normalize(tangent * lw.x + bitangent * lw.y + n * lw.z);
    s.f = (refl ? P.r : P.t) * (1.0f / PI_F);
    s.pdf = (refl ? prob_r : (1.0f - prob_r)) * ct / PI_F;
    s.is_specular = false, s.eta_scale = 1.0f;
    return s;
}
inline Spec dt_eval(const DTParams& P, V3 wo_world, V3 wi_world, V3 n, float& pdf) {
    pdf = 0.0f;
    float ci = dot(wi_world, n), co = dot(wo_world, n);
    float aci = std::fabs(ci);
    if (aci < 1e-6f) return Spec();
    if (P.pr + P.pt < 1e-10f) return Spec();
    if (ci * co > 0.0f) {
        pdf = (P.pr / (P.pr + P.pt)) * aci / PI_F;
        return P.r * (1.0f / PI_F);
    }
    pdf = (P.pt / (P.pr + P.pt)) * aci / PI_F;
    return P.t * (1.0f / PI_F);
}

// material-dispatch.jl:23-53 over every material kind: the kinds of this header, else hko_bsdf.h
inline BSDFSample sample_bsdf_all(const MaterialCtx& c, int32_t mat_idx, V3 wo, V3 n, const TexCtx& uv, const Wavelengths& w, V2 u, float rng, bool regularize) {
    const hk_material& m = c.materials[mat_idx];
    switch (m.kind) {
        case HK_MAT_COATED_DIFFUSE:
        case HK_MAT_COATED_DIFFUSE_TRANSMISSION: return layered_sample(layered_params(c, m, uv, w, regularize), wo, n, u, rng);
        case HK_MAT_COATED_CONDUCTOR: return cc_sample(cc_params(c, m, uv, w, regularize), wo, n, u, rng);
        case HK_MAT_THIN_DIELECTRIC: return thin_dielectric_sample(eval_tex(c.textures, m.f[0], uv), wo, n, rng);
        case HK_MAT_DIFFUSE_TRANSMISSION: return dt_sample(dt_params(c, m, uv, w), wo, n, u, rng);
        default: return sample_bsdf(c, mat_idx, wo, n, uv, w, u, rng, regularize);
    }
}
inline Spec eval_bsdf_all(const MaterialCtx& c, int32_t mat_idx, V3 wo, V3 wi, V3 n, const TexCtx& uv, const Wavelengths& w, float& pdf) {
    const hk_material& m = c.materials[mat_idx];
    switch (m.kind) {
        case HK_MAT_COATED_DIFFUSE:
        case HK_MAT_COATED_DIFFUSE_TRANSMISSION: return layered_eval(layered_params(c, m, uv, w, false), wo, wi, n, pdf);
        case HK_MAT_COATED_CONDUCTOR: return cc_eval(cc_params(c, m, uv, w, false), wo, wi, n, pdf);
        case HK_MAT_THIN_DIELECTRIC: pdf = 0.0f; return Spec();
        case HK_MAT_DIFFUSE_TRANSMISSION: return dt_eval(dt_params(c, m, uv, w), wo, wi, n, pdf);
        default: return eval_bsdf(c, mat_idx, wo, wi, n, uv, w, pdf);
    }
}

}  // namespace hko

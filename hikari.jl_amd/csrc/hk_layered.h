// hk_layered.h — device code for the layered / thin / transmissive material kinds.
// Included by hk_device.h (after the microfacet / Fresnel helpers, before sample_bsdf<KIND>).
//
// Reference: src/materials/spectral-eval.jl
//   layer_transmittance :836-839, sample_hg_phase_spectral :846-873, hg_phase_pdf :880-884,
//   sample_dielectric_interface :965-1046, refract_pbrt :1055-1077, refract_microfacet :1084-1104,
//   diffuse bottom :1143-1198, eval/pdf_dielectric_interface :1449-1575,
//   CoatedDiffuse sample :1233-1441 / eval :1563-1832 / pdf :1840-1937,
//   ThinDielectric :1975-2051, DiffuseTransmission :2083-2218, diffuse-transmission bottom :2253-2337,
//   CoatedDiffuseTransmission :2341-2832, CoatedConductor :2877-3420.
//
// One stochastic walk serves both coated kinds: the bottom interface is a compile-time switch (DT) between
// the DiffuseBxDF and the DiffuseTransmissionBxDF restatements, whose random-number consumption differs.
// The walk RNG is a PCG32 seeded from the bit patterns of the local directions, exactly as the reference does,
// so a walk reproduces bit-for-bit whenever wo_local / wi_local do.
#pragma once

namespace hkd {

enum { BX_R = 1, BX_T = 2, BX_ALL = 3 };

struct LSample {
    S4 f;
    v3 wi;
    float pdf;
    float eta;
    bool is_reflection, is_specular, valid;
};
HKD LSample l_invalid() { return LSample{s4(0.0f), mk3(0, 0, 0), 0.0f, 1.0f, false, false, false}; }
HKD LSample l_make(S4 f, v3 wi, float pdf, bool refl, bool spec, float eta) { return LSample{f, wi, pdf, eta, refl, spec, true}; }
HKD bool l_dead(const LSample& s) { return !s.valid || s.pdf == 0.0f || s.wi.z == 0.0f; }

HKD float layer_tr(float thickness, v3 w) {
    if (fabsf(thickness) <= 1.1920929e-7f) return 1.0f;
    return expf(-fabsf(thickness / w.z));
}
HKD float lhg_pdf(float g, float ct) {
    float g2 = g * g;
    float denom = 1.0f + g2 - 2.0f * g * ct;
    return (1.0f - g2) / (4.0f * PI_F * denom * sqrtf(maxf(1e-10f, denom)));
}
HKD v3 lhg_sample(float g, v3 wo, v2 u, float& p) {
    float ct;
    if (fabsf(g) < 1e-3f)
        ct = 1.0f - 2.0f * u.x;
    else {
        float g2 = g * g;
        float sq = (1.0f - g2) / (1.0f - g + 2.0f * g * u.x);
        ct = clampf((1.0f + g2 - sq * sq) / (2.0f * g), -1.0f, 1.0f);
    }
    float st = sqrtf(maxf(0.0f, 1.0f - ct * ct));
    float phi = 2.0f * PI_F * u.y;
    v3 t1, t2;
    coordinate_system(-wo, t1, t2);
    float sphi, cphi;
    jl_sincos(phi, sphi, cphi);
    v3 wi = st * cphi * t1 + st * sphi * t2 + ct * (-wo);
    wi = normalize(wi);
    p = lhg_pdf(g, ct);
    return wi;
}
HKD float sample_exponential(float u, float a) { return -logf(1.0f - u) / a; }
HKD float power_heuristic1(float fp, float gp) {
    float f2 = fp * fp, g2 = gp * gp;
    if (f2 + g2 == 0.0f) return 0.0f;
    return f2 / (f2 + g2);
}
HKD bool same_hemi(v3 a, v3 b) { return a.z * b.z > 0.0f; }

HKD bool refract_pbrt(v3 wo, float eta, v3& wi, float& etap) {
    float ci = wo.z;
    etap = ci > 0.0f ? eta : (1.0f / eta);
    float s2i = maxf(0.0f, 1.0f - ci * ci);
    float s2t = s2i / (etap * etap);
    if (s2t >= 1.0f) return false;
    float ct = sqrtf(1.0f - s2t);
    float cts = ci > 0.0f ? -ct : ct;
    wi = normalize(mk3(-wo.x / etap, -wo.y / etap, cts));
    return true;
}
HKD bool refract_microfacet(v3 wo, v3 wm, float eta, v3& wi, float& etap) {
    float ci = dot(wo, wm);
    etap = ci > 0.0f ? eta : (1.0f / eta);
    float s2i = maxf(0.0f, 1.0f - ci * ci);
    float s2t = s2i / (etap * etap);
    if (s2t >= 1.0f) return false;
    float ct = sqrtf(1.0f - s2t);
    float cts = ci > 0.0f ? -ct : ct;
    wi = normalize(-wo / etap + (ci / etap + cts) * wm);
    return true;
}

// ---- top interface: DielectricBxDF ---------------------------------------------------------------
struct Coat {
    float ax, ay, eta;
    bool smooth;  // trowbridge_reitz_effectively_smooth(ax, ay) || eta == 1
};
HKD LSample sample_dielectric_interface(v3 wo, float uc, v2 u, const Coat& C, int flags) {
    if (C.smooth) {
        float R = fresnel_dielectric(wo.z, C.eta), T = 1.0f - R;
        float pr = (flags & BX_R) ? R : 0.0f, pt = (flags & BX_T) ? T : 0.0f;
        if (pr == 0.0f && pt == 0.0f) return l_invalid();
        if (uc < pr / (pr + pt)) {
            v3 wi = mk3(-wo.x, -wo.y, wo.z);
            return l_make(s4(R / fabsf(wi.z)), wi, pr / (pr + pt), true, true, 1.0f);
        }
        v3 wi;
        float etap;
        if (!refract_pbrt(wo, C.eta, wi, etap)) return l_invalid();
        return l_make(s4(T / fabsf(wi.z)), wi, pt / (pr + pt), false, true, etap);
    }
    v3 wm = tr_sample_wm(wo, u, C.ax, C.ay);
    float com = dot(wo, wm);
    float R = fresnel_dielectric(com, C.eta), T = 1.0f - R;
    float pr = (flags & BX_R) ? R : 0.0f, pt = (flags & BX_T) ? T : 0.0f;
    if (pr == 0.0f && pt == 0.0f) return l_invalid();
    if (uc < pr / (pr + pt)) {
        v3 wi = reflect(wo, wm);
        if (!same_hemi(wo, wi)) return l_invalid();
        float pdf = tr_pdf(wo, wm, C.ax, C.ay) / (4.0f * fabsf(com)) * pr / (pr + pt);
        float D = tr_d(wm, C.ax, C.ay), G = tr_g(wo, wi, C.ax, C.ay);
        float f = D * G * R / (4.0f * wo.z * wi.z);
        return l_make(s4(f), wi, pdf, true, false, 1.0f);
    }
    v3 wi;
    float etap;
    if (!refract_microfacet(wo, wm, C.eta, wi, etap) || same_hemi(wo, wi) || wi.z == 0.0f) return l_invalid();
    float s = dot(wi, wm) + dot(wo, wm) / etap;
    float denom = s * s;
    float dwm_dwi = fabsf(dot(wi, wm)) / denom;
    float pdf = tr_pdf(wo, wm, C.ax, C.ay) * dwm_dwi * pt / (pr + pt);
    float D = tr_d(wm, C.ax, C.ay), G = tr_g(wo, wi, C.ax, C.ay);
    float f = T * D * G * fabsf(dot(wi, wm) * dot(wo, wm) / (wi.z * wo.z * denom));
    return l_make(s4(f), wi, pdf, false, false, etap);
}
HKD float eval_dielectric_interface(v3 wo, v3 wi, const Coat& C) {  // f only: the reference discards this pdf
    if (C.smooth) return 0.0f;
    if (same_hemi(wo, wi)) {
        v3 wh = normalize(wo + wi);
        if (wh.z < 0.0f) wh = -wh;
        float R = fresnel_dielectric(dot(wo, wh), C.eta);
        float D = tr_d(wh, C.ax, C.ay), G = tr_g(wo, wi, C.ax, C.ay);
        return D * G * R / (4.0f * wo.z * wi.z);
    }
    float etap = wo.z > 0.0f ? C.eta : (1.0f / C.eta);
    v3 wh = normalize(wo + wi * etap);
    if (wh.z < 0.0f) wh = -wh;
    float coh = dot(wo, wh), cih = dot(wi, wh);
    if (coh * cih > 0.0f) return 0.0f;
    float R = fresnel_dielectric(coh, C.eta), T = 1.0f - R;
    float s = cih + coh / etap;
    float denom = s * s;
    float D = tr_d(wh, C.ax, C.ay), G = tr_g(wo, wi, C.ax, C.ay);
    return T * D * G * fabsf(cih * coh / (wo.z * wi.z * denom));
}
HKD float pdf_dielectric_interface(v3 wo, v3 wi, const Coat& C, int flags) {
    if (C.smooth) return 0.0f;
    if (same_hemi(wo, wi)) {
        if (!(flags & BX_R)) return 0.0f;
        v3 wh = normalize(wo + wi);
        if (wh.z < 0.0f) wh = -wh;
        float coh = fabsf(dot(wo, wh));
        float R = fresnel_dielectric(coh, C.eta), T = 1.0f - R;
        float pr = (flags & BX_R) ? R : 0.0f, pt = (flags & BX_T) ? T : 0.0f;
        float pdf = tr_pdf(wo, wh, C.ax, C.ay) / (4.0f * coh);
        return pdf * pr / (pr + pt);
    }
    if (!(flags & BX_T)) return 0.0f;
    float etap = wo.z > 0.0f ? C.eta : (1.0f / C.eta);
    v3 wh = normalize(wo + wi * etap);
    if (wh.z < 0.0f) wh = -wh;
    float coh = dot(wo, wh), cih = dot(wi, wh);
    if (coh * cih > 0.0f) return 0.0f;
    float R = fresnel_dielectric(fabsf(coh), C.eta), T = 1.0f - R;
    float pr = (flags & BX_R) ? R : 0.0f, pt = (flags & BX_T) ? T : 0.0f;
    float s = cih + coh / etap;
    float denom = s * s;
    float dwm_dwi = fabsf(cih) / denom;
    float pdf = tr_pdf(wo, wh, C.ax, C.ay) * dwm_dwi;
    return pdf * pt / (pr + pt);
}

// ---- bottom interface ------------------------------------------------------------------------------
struct Bottom {
    S4 refl, trans;
    float pr_max, pt_max;
};
template <bool DT>
HKD LSample bottom_sample(const Bottom& b, v3 wo, v2 u, float uc, int flags) {
    if (!DT) {
        if (!(flags & BX_R)) return l_invalid();
        v3 wi = cosine_sample_hemisphere(u);
        if (wo.z < 0.0f) wi = mk3(wi.x, wi.y, -wi.z);
        float ci = fabsf(wi.z);
        if (ci < 1e-6f) return l_invalid();
        return l_make(b.refl * (1.0f / PI_F), wi, ci / PI_F, true, false, 1.0f);
    }
    float pr = (flags & BX_R) ? b.pr_max : 0.0f, pt = (flags & BX_T) ? b.pt_max : 0.0f;
    if (pr + pt < 1e-10f) return l_invalid();
    float prob_r = pr / (pr + pt);
    v3 wi = cosine_sample_hemisphere(u);
    bool refl = uc < prob_r;
    if (refl ? (wo.z < 0.0f) : (wo.z > 0.0f)) wi = mk3(wi.x, wi.y, -wi.z);
    float ci = fabsf(wi.z);
    if (ci < 1e-6f) return l_invalid();
    if (refl) return l_make(b.refl * (1.0f / PI_F), wi, prob_r * ci / PI_F, true, false, 1.0f);
    return l_make(b.trans * (1.0f / PI_F), wi, (1.0f - prob_r) * ci / PI_F, false, false, 1.0f);
}
template <bool DT>
HKD S4 bottom_eval(const Bottom& b, v3 wo, v3 wi, float& pdf) {
    pdf = 0.0f;
    if (!DT) {
        if (!same_hemi(wo, wi)) return s4(0.0f);
        pdf = fabsf(wi.z) / PI_F;
        return b.refl * (1.0f / PI_F);
    }
    if (b.pr_max + b.pt_max < 1e-10f) return s4(0.0f);
    float aci = fabsf(wi.z);
    if (same_hemi(wo, wi)) {
        pdf = (b.pr_max / (b.pr_max + b.pt_max)) * aci / PI_F;
        return b.refl * (1.0f / PI_F);
    }
    pdf = (b.pt_max / (b.pr_max + b.pt_max)) * aci / PI_F;
    return b.trans * (1.0f / PI_F);
}
template <bool DT>
HKD float bottom_pdf(const Bottom& b, v3 wo, v3 wi, int flags) {
    if (!DT) {
        if (!same_hemi(wo, wi)) return 0.0f;
        return fabsf(wi.z) / PI_F;
    }
    float pr = (flags & BX_R) ? b.pr_max : 0.0f, pt = (flags & BX_T) ? b.pt_max : 0.0f;
    if (pr + pt < 1e-10f) return 0.0f;
    float aci = fabsf(wi.z);
    if (same_hemi(wo, wi)) return (pr / (pr + pt)) * aci / PI_F;
    return (pt / (pr + pt)) * aci / PI_F;
}

struct LayeredParams {
    Bottom bottom;
    Coat coat;
    S4 albedo;
    float thickness, g;
    int max_depth, n_samples;
    bool has_medium, is_smooth;  // is_smooth: effectively-smooth test alone (without the eta == 1 clause)
};

// raw (r,g,b) of a colour parameter: constant or bilinear texture value
HKD void rgb_param_raw(const DScene& sc, const DSpectrumParam& p, const TexCtx& uv, float o[4]) {
    if (p.tex < 0) {
        o[0] = p.rgba[0], o[1] = p.rgba[1], o[2] = p.rgba[2], o[3] = p.rgba[3];
        return;
    }
    o[0] = o[1] = o[2] = 0.0f, o[3] = 1.0f;
    tex_bilinear(sc.textures[p.tex], uv, o);
}
// bounded uplift of a parameter whose constant case the host baked (uplift_rgb clamps to [0,1] itself)
HKD S4 param_bounded(const DScene& sc, const DTables& T, const DSpectrumParam& p, const float raw[4], S4 lambda) {
    return eval_bounded(p.tex < 0 ? p.coef : coef_bounded(T, raw[0], raw[1], raw[2]), lambda);
}

template <bool DT>
HKD LayeredParams layered_params(const DScene& sc, const DTables& T, const DMaterial& m, const TexCtx& uv, S4 lambda, bool regularize) {
    LayeredParams P;
    float raw[4];
    rgb_param_raw(sc, m.rgb[0], uv, raw);
    P.bottom.refl = param_bounded(sc, T, m.rgb[0], raw, lambda);
    P.bottom.trans = s4(0.0f);
    P.bottom.pr_max = P.bottom.pt_max = 0.0f;
    if (DT) {
        P.bottom.pr_max = maxf(maxf(clampf(raw[0], 0.0f, 1.0f), clampf(raw[1], 0.0f, 1.0f)), clampf(raw[2], 0.0f, 1.0f));
        rgb_param_raw(sc, m.rgb[1], uv, raw);
        P.bottom.trans = param_bounded(sc, T, m.rgb[1], raw, lambda);
        P.bottom.pt_max = maxf(maxf(clampf(raw[0], 0.0f, 1.0f), clampf(raw[1], 0.0f, 1.0f)), clampf(raw[2], 0.0f, 1.0f));
    }
    const DSpectrumParam& alb = m.rgb[DT ? 2 : 1];
    rgb_param_raw(sc, alb, uv, raw);
    P.has_medium = !(raw[0] == 0.0f && raw[1] == 0.0f && raw[2] == 0.0f);
    P.albedo = P.has_medium ? param_bounded(sc, T, alb, raw, lambda) : s4(0.0f);
    float ax = eval_f32(sc, m, 0, uv), ay = eval_f32(sc, m, 1, uv);
    if (m.flags & HK_MATF_REMAP_ROUGHNESS) {
        ax = sqrtf(ax);
        ay = sqrtf(ay);
    }
    if (regularize) {
        ax = ax < 0.3f ? clampf(2.0f * ax, 0.1f, 0.3f) : ax;
        ay = ay < 0.3f ? clampf(2.0f * ay, 0.1f, 0.3f) : ay;
    }
    P.coat.ax = ax;
    P.coat.ay = ay;
    P.coat.eta = eval_f32(sc, m, 3, uv);
    P.is_smooth = tr_smooth(ax, ay);
    P.coat.smooth = P.is_smooth || P.coat.eta == 1.0f;
    P.thickness = maxf(eval_f32(sc, m, 2, uv), 1.1920929e-7f);
    P.g = clampf(eval_f32(sc, m, 4, uv), -0.99f, 0.99f);
    P.max_depth = m.i[0];
    P.n_samples = m.i[1];
    return P;
}

// LayeredBxDF::Sample_f as restated by the reference (:1233-1441 / :2341-2497)
template <bool DT>
HKD BSDFSample layered_sample(const LayeredParams& P, v3 wo_w, v3 n, v2 sample_u, float rng_in) {
    float wdn = dot(wo_w, n);
    if (fabsf(wdn) < 1e-6f) return invalid_sample();
    v3 t, b;
    coordinate_system(n, t, b);
    v3 wo = mk3(dot(wo_w, t), dot(wo_w, b), wdn);
    bool flip = wo.z < 0.0f;
    if (flip) wo = -wo;
    LSample bs = sample_dielectric_interface(wo, rng_in, sample_u, P.coat, BX_ALL);
    if (l_dead(bs)) return invalid_sample();
    BSDFSample out;
    if (bs.is_reflection) {
        v3 wl = flip ? -bs.wi : bs.wi;
        out.wi = normalize(t * wl.x + b * wl.y + n * wl.z);
        out.f = bs.f, out.pdf = bs.pdf, out.is_specular = bs.is_specular, out.eta_scale = 1.0f;
        return out;
    }
    v3 w = bs.wi;
    bool specular_path = bs.is_specular;
    S4 f = bs.f * fabsf(w.z);
    float pdf = bs.pdf;
    float z = P.thickness;
    PCG32 rng = pcg32_init(pbrt_hash((uint64_t)0, wo), pbrt_hash(rng_in, sample_u));
    for (int depth = 0; depth < P.max_depth; ++depth) {
        float rr_beta = max_component(f) / pdf;
        if (depth > 3 && rr_beta < 0.25f) {
            float q = maxf(0.0f, 1.0f - rr_beta);
            if (pcg32_f32(rng) < q) return invalid_sample();
            pdf *= 1.0f - q;
        }
        if (w.z == 0.0f) return invalid_sample();
        if (P.has_medium) {
            float dz = sample_exponential(pcg32_f32(rng), 1.0f / fabsf(w.z));
            float zp = w.z > 0.0f ? (z + dz) : (z - dz);
            if (zp == z) return invalid_sample();
            if (0.0f < zp && zp < P.thickness) {
                float u1 = pcg32_f32(rng), u2 = pcg32_f32(rng);
                float pp;
                v3 wp = lhg_sample(P.g, -w, mk2(u1, u2), pp);
                if (pp == 0.0f || wp.z == 0.0f) return invalid_sample();
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
        float uc = pcg32_f32(rng), u1 = pcg32_f32(rng), u2 = pcg32_f32(rng);
        LSample bi = (z == 0.0f) ? bottom_sample<DT>(P.bottom, -w, mk2(u1, u2), uc, BX_ALL) : sample_dielectric_interface(-w, uc, mk2(u1, u2), P.coat, BX_ALL);
        if (l_dead(bi)) return invalid_sample();
        f = f * bi.f;
        pdf *= bi.pdf;
        specular_path = specular_path && bi.is_specular;
        w = bi.wi;
        if (!bi.is_reflection) {
            v3 wl = flip ? -w : w;
            out.wi = normalize(t * wl.x + b * wl.y + n * wl.z);
            out.f = f, out.pdf = pdf, out.is_specular = specular_path, out.eta_scale = bi.eta;
            return out;
        }
        f = f * fabsf(bi.wi.z);
    }
    return invalid_sample();
}

// pdf_layered_bsdf :1840-1937 / pdf_layered_bsdf_dt :2748-2832
template <bool DT>
HKD float layered_pdf(const LayeredParams& P, v3 wo, v3 wi) {
    PCG32 rng = pcg32_init(pbrt_hash((uint64_t)0, wi), pbrt_hash(wo));
    const bool sh = same_hemi(wo, wi);
    const Bottom& B = P.bottom;
    float pdf_sum = 0.0f;
    if (sh && !P.is_smooth) pdf_sum += (float)P.n_samples * pdf_dielectric_interface(wo, wi, P.coat, BX_R);
    for (int s = 0; s < P.n_samples; ++s) {
        float uc1 = pcg32_f32(rng), u1 = pcg32_f32(rng), u2 = pcg32_f32(rng);
        LSample wos = sample_dielectric_interface(wo, uc1, mk2(u1, u2), P.coat, BX_T);
        if (sh) {
            float uc2 = pcg32_f32(rng), u3 = pcg32_f32(rng), u4 = pcg32_f32(rng);
            LSample wis = sample_dielectric_interface(wi, uc2, mk2(u3, u4), P.coat, BX_T);
            if (wos.valid && wos.pdf > 0.0f && wis.valid && wis.pdf > 0.0f) {
                if (P.is_smooth)
                    pdf_sum += bottom_pdf<DT>(B, -wos.wi, -wis.wi, BX_ALL);
                else {
                    float u5 = pcg32_f32(rng), u6 = pcg32_f32(rng);
                    float uc3 = DT ? pcg32_f32(rng) : 0.0f;
                    LSample rs = bottom_sample<DT>(B, -wos.wi, mk2(u5, u6), uc3, BX_ALL);
                    if (rs.valid && rs.pdf > 0.0f) {
                        float r_pdf = bottom_pdf<DT>(B, -wos.wi, -wis.wi, BX_ALL);
                        pdf_sum += power_heuristic1(wis.pdf, r_pdf) * r_pdf;
                        float t_pdf = pdf_dielectric_interface(-rs.wi, wi, P.coat, BX_ALL);
                        pdf_sum += power_heuristic1(rs.pdf, t_pdf) * t_pdf;
                    }
                }
            }
        } else {
            if (!wos.valid || wos.pdf == 0.0f || wos.is_reflection) continue;
            float uc2 = DT ? pcg32_f32(rng) : 0.0f;
            float u3 = pcg32_f32(rng), u4 = pcg32_f32(rng);
            LSample wis = bottom_sample<DT>(B, wi, mk2(u3, u4), uc2, BX_T);
            if (!wis.valid || wis.pdf == 0.0f || wis.is_reflection) continue;
            if (P.is_smooth)
                pdf_sum += bottom_pdf<DT>(B, -wos.wi, wi, BX_ALL);
            else
                pdf_sum += (pdf_dielectric_interface(wo, -wis.wi, P.coat, BX_ALL) + bottom_pdf<DT>(B, -wos.wi, wi, BX_ALL)) / 2.0f;
        }
    }
    // quirk Q26: lerp(v1, v2, t) called with pbrt's (t, a, b) order => (1 - p)*0.9 + p/(4 pi)
    return lerpf(0.9f, 1.0f / (4.0f * PI_F), pdf_sum / (float)P.n_samples);
}

// LayeredBxDF::f as restated by the reference (:1563-1832 / :2501-2744)
template <bool DT>
HKD S4 layered_eval(const LayeredParams& P, v3 wo_w, v3 wi_w, v3 n, float& pdf) {
    pdf = 0.0f;
    const Bottom& B = P.bottom;
    v3 t, b;
    coordinate_system(n, t, b);
    v3 wo = mk3(dot(wo_w, t), dot(wo_w, b), dot(wo_w, n));
    v3 wi = mk3(dot(wi_w, t), dot(wi_w, b), dot(wi_w, n));
    if (wo.z < 0.0f) {
        wo = -wo;
        wi = -wi;
    }
    if (fabsf(wo.z) < 1e-6f || fabsf(wi.z) < 1e-6f) return s4(0.0f);
    const bool sh = same_hemi(wo, wi);
    const bool exit_at_bottom = !sh;
    const float exit_z = exit_at_bottom ? 0.0f : P.thickness;
    S4 fr = s4(0.0f);
    if (sh) fr = fr + s4(eval_dielectric_interface(wo, wi, P.coat)) * (float)P.n_samples;
    PCG32 rng = pcg32_init(pbrt_hash((uint64_t)0, wo), pbrt_hash(wi));
    const bool is_smooth = P.is_smooth;
    for (int s = 0; s < P.n_samples; ++s) {
        float uc = pcg32_f32(rng), u1 = pcg32_f32(rng), u2 = pcg32_f32(rng);
        LSample wos = sample_dielectric_interface(wo, uc, mk2(u1, u2), P.coat, BX_T);
        if (l_dead(wos)) continue;
        uc = pcg32_f32(rng), u1 = pcg32_f32(rng), u2 = pcg32_f32(rng);
        LSample wis = exit_at_bottom ? bottom_sample<DT>(B, wi, mk2(u1, u2), uc, BX_T) : sample_dielectric_interface(wi, uc, mk2(u1, u2), P.coat, BX_T);
        if (l_dead(wis)) continue;
        S4 beta = wos.f * fabsf(wos.wi.z) / wos.pdf;
        float z = P.thickness;
        v3 w = wos.wi;
        for (int depth = 0; depth < P.max_depth; ++depth) {
            if (depth > 3 && max_component(beta) < 0.25f) {
                float q = maxf(0.0f, 1.0f - max_component(beta));
                if (pcg32_f32(rng) < q) break;
                beta = beta / (1.0f - q);
            }
            if (P.has_medium) {
                float dz = sample_exponential(pcg32_f32(rng), 1.0f / fabsf(w.z));
                float zp = w.z > 0.0f ? (z + dz) : (z - dz);
                if (zp == z) continue;
                if (0.0f < zp && zp < P.thickness) {
                    float phase_val = lhg_pdf(P.g, dot(-w, -wis.wi));
                    float wt = 1.0f;
                    if (exit_at_bottom || !is_smooth) wt = power_heuristic1(wis.pdf, phase_val);
                    fr = fr + beta * P.albedo * phase_val * wt * layer_tr(zp - exit_z, wis.wi) * wis.f / wis.pdf;
                    float pu1 = pcg32_f32(rng), pu2 = pcg32_f32(rng);
                    float pp;
                    v3 wp = lhg_sample(P.g, -w, mk2(pu1, pu2), pp);
                    if (pp == 0.0f || wp.z == 0.0f) break;
                    beta = beta * P.albedo * pp / pp;
                    w = wp;
                    z = zp;
                    if ((z < exit_z && w.z > 0.0f) || (z > exit_z && w.z < 0.0f)) {
                        S4 fe;
                        float epdf;
                        if (exit_at_bottom)
                            fe = bottom_eval<DT>(B, -w, wi, epdf);
                        else {
                            if (is_smooth) continue;
                            fe = s4(eval_dielectric_interface(-w, wi, P.coat));
                            epdf = pdf_dielectric_interface(-w, wi, P.coat, BX_T);
                        }
                        if (max_component(fe) > 0.0f) fr = fr + beta * layer_tr(zp - exit_z, wp) * fe * power_heuristic1(pp, epdf);
                    }
                    continue;
                }
                z = clampf(zp, 0.0f, P.thickness);
            } else {
                z = (z == P.thickness) ? 0.0f : P.thickness;
                beta = beta * layer_tr(P.thickness, w);
            }
            if (z == exit_z) {
                uc = pcg32_f32(rng), u1 = pcg32_f32(rng), u2 = pcg32_f32(rng);
                LSample b2 = exit_at_bottom ? bottom_sample<DT>(B, -w, mk2(u1, u2), uc, BX_R) : sample_dielectric_interface(-w, uc, mk2(u1, u2), P.coat, BX_R);
                if (l_dead(b2)) break;
                beta = beta * b2.f * fabsf(b2.wi.z) / b2.pdf;
                w = b2.wi;
            } else {
                const bool ne_bottom = (z == 0.0f);
                const bool ne_specular = !ne_bottom && is_smooth;
                if (!ne_specular) {
                    float dummy;
                    S4 f_nee = ne_bottom ? bottom_eval<DT>(B, -w, -wis.wi, dummy) : s4(eval_dielectric_interface(-w, -wis.wi, P.coat));
                    if (max_component(f_nee) > 0.0f) {
                        float wt = 1.0f;
                        if (!exit_at_bottom || !is_smooth) {
                            float nee_pdf = ne_bottom ? bottom_pdf<DT>(B, -w, -wis.wi, BX_ALL) : pdf_dielectric_interface(-w, -wis.wi, P.coat, BX_ALL);
                            wt = power_heuristic1(wis.pdf, nee_pdf);
                        }
                        fr = fr + beta * f_nee * fabsf(wis.wi.z) * wt * layer_tr(P.thickness, wis.wi) * wis.f / wis.pdf;
                    }
                }
                uc = pcg32_f32(rng), u1 = pcg32_f32(rng), u2 = pcg32_f32(rng);
                LSample b2 = ne_bottom ? bottom_sample<DT>(B, -w, mk2(u1, u2), uc, BX_R) : sample_dielectric_interface(-w, uc, mk2(u1, u2), P.coat, BX_R);
                if (l_dead(b2)) break;
                beta = beta * b2.f * fabsf(b2.wi.z) / b2.pdf;
                w = b2.wi;
                if (!is_smooth || exit_at_bottom) {
                    float dummy;
                    S4 fe = exit_at_bottom ? bottom_eval<DT>(B, -w, wi, dummy) : s4(eval_dielectric_interface(-w, wi, P.coat));
                    if (max_component(fe) > 0.0f) {
                        float wt3 = 1.0f;
                        if (!ne_specular) {
                            float epdf = exit_at_bottom ? bottom_pdf<DT>(B, -w, wi, BX_ALL) : pdf_dielectric_interface(-w, wi, P.coat, BX_T);
                            wt3 = power_heuristic1(b2.pdf, epdf);
                        }
                        fr = fr + beta * layer_tr(P.thickness, b2.wi) * fe * wt3;
                    }
                }
            }
        }
    }
    fr = fr / (float)P.n_samples;
    pdf = layered_pdf<DT>(P, wo, wi);
    return fr;
}

// ---- CoatedConductor: the reference's analytic two-lobe form (:2877-3420) ---------------------------
struct CCParams {
    S4 ce, ck, albedo;
    float ieta, iax, iay, cax, cay, thickness;
    bool has_medium;
};
HKD CCParams cc_params(const DScene& sc, const DTables& T, const DMaterial& m, const TexCtx& uv, S4 lambda, bool regularize) {
    CCParams P;
    P.ieta = eval_f32(sc, m, 2, uv);
    if (P.ieta == 0.0f) P.ieta = 1.0f;
    P.iax = eval_f32(sc, m, 0, uv), P.iay = eval_f32(sc, m, 1, uv);
    P.cax = eval_f32(sc, m, 3, uv), P.cay = eval_f32(sc, m, 4, uv);
    if (m.flags & HK_MATF_REMAP_ROUGHNESS) {
        P.iax = sqrtf(P.iax), P.iay = sqrtf(P.iay);
        P.cax = sqrtf(P.cax), P.cay = sqrtf(P.cay);
    }
    if (regularize) {
        P.iax = P.iax < 0.3f ? clampf(2.0f * P.iax, 0.1f, 0.3f) : P.iax;
        P.iay = P.iay < 0.3f ? clampf(2.0f * P.iay, 0.1f, 0.3f) : P.iay;
        P.cax = P.cax < 0.3f ? clampf(2.0f * P.cax, 0.1f, 0.3f) : P.cax;
        P.cay = P.cay < 0.3f ? clampf(2.0f * P.cay, 0.1f, 0.3f) : P.cay;
    }
    if (m.flags & HK_MATF_USE_ETA_K) {
        P.ce = eval_ior(sc, T, m, 0, uv, lambda);
        P.ck = eval_ior(sc, T, m, 1, uv, lambda);
    } else {
        // reflectance mode: the host bakes clamp(r, 0, 0.9999) for a constant colour
        float raw[4];
        rgb_param_raw(sc, m.rgb[2], uv, raw);
        S4 rs = eval_bounded(m.rgb[2].tex < 0 ? m.rgb[2].coef
                                              : coef_bounded(T, clampf(raw[0], 0.0f, 0.9999f), clampf(raw[1], 0.0f, 0.9999f), clampf(raw[2], 0.0f, 0.9999f)),
                             lambda);
        P.ce = s4(1.0f);
        S4 om = s4(1.0f) - rs;
        om = s4(maxf(0.0f, om.x), maxf(0.0f, om.y), maxf(0.0f, om.z), maxf(0.0f, om.w)) + s4(1e-6f);
        P.ck = (2.0f * s4(sqrtf(rs.x), sqrtf(rs.y), sqrtf(rs.z), sqrtf(rs.w))) / s4(sqrtf(om.x), sqrtf(om.y), sqrtf(om.z), sqrtf(om.w));
    }
    P.ce = P.ce / P.ieta;
    P.ck = P.ck / P.ieta;
    P.thickness = maxf(eval_f32(sc, m, 5, uv), 1.1920929e-7f);
    float raw[4];
    rgb_param_raw(sc, m.rgb[3], uv, raw);
    P.has_medium = !(raw[0] == 0.0f && raw[1] == 0.0f && raw[2] == 0.0f);
    P.albedo = P.has_medium ? param_bounded(sc, T, m.rgb[3], raw, lambda) : s4(0.0f);
    return P;
}
HKD BSDFSample cc_sample(const CCParams& P, v3 wo_w, v3 n, v2 sample_u, float rng) {
    float wdn = dot(wo_w, n);
    if (fabsf(wdn) < 1e-6f) return invalid_sample();
    v3 t, b;
    coordinate_system(n, t, b);
    v3 wo = mk3(dot(wo_w, t), dot(wo_w, b), wdn);
    bool flip = wo.z < 0.0f;
    if (flip) wo = -wo;
    float cos_o = fabsf(wo.z);
    bool i_smooth = tr_smooth(P.iax, P.iay), c_smooth = tr_smooth(P.cax, P.cay);
    BSDFSample out;
    out.eta_scale = 1.0f;
    out.is_specular = false;
    v3 wl;  // outgoing local direction (before the two-sided flip)
    if (i_smooth) {
        float Fi = fresnel_dielectric(cos_o, P.ieta);
        if (rng < Fi) {
            wl = mk3(-wo.x, -wo.y, wo.z);
            out.f = s4(1.0f), out.pdf = 1.0f, out.is_specular = true;
        } else {
            float s2t = maxf(0.0f, 1.0f - cos_o * cos_o) / (P.ieta * P.ieta);
            if (s2t >= 1.0f) return invalid_sample();
            float ct_in = sqrtf(1.0f - s2t);
            float T_in = 1.0f - Fi;
            if (c_smooth) {
                v3 wb = normalize(mk3(-wo.x / P.ieta, -wo.y / P.ieta, ct_in));
                S4 Fc = fr_complex4(ct_in, P.ce, P.ck);
                float s2o = maxf(0.0f, 1.0f - wb.z * wb.z) * (P.ieta * P.ieta);
                if (s2o >= 1.0f) return invalid_sample();
                float c_out = sqrtf(1.0f - s2o);
                float T_out = 1.0f - fresnel_dielectric(c_out, P.ieta);
                S4 ltr = s4(1.0f);
                if (P.has_medium) {
                    float tr = layer_tr(P.thickness, mk3(0, 0, ct_in));
                    ltr = tr * tr * P.albedo;
                }
                wl = mk3(-wo.x, -wo.y, wo.z);
                out.f = Fc * T_in * T_out * ltr / cos_o;
                out.pdf = 1.0f - Fi, out.is_specular = true;
            } else {
                v3 woc = normalize(mk3(wo.x / P.ieta, wo.y / P.ieta, ct_in));
                float cax = maxf(P.cax, 1e-4f), cay = maxf(P.cay, 1e-4f);
                v3 wm = tr_sample_wm(woc, sample_u, cax, cay);
                float com = dot(woc, wm);
                if (com < 0.0f) return invalid_sample();
                v3 wic = -woc + 2.0f * com * wm;
                if (wic.z < 0.0f) return invalid_sample();
                S4 Fc = fr_complex4(fabsf(com), P.ce, P.ck);
                float D = tr_d(wm, cax, cay), G = tr_g(woc, wic, cax, cay);
                S4 fc = D * Fc * G / (4.0f * fabsf(woc.z) * fabsf(wic.z));
                float s2o = (wic.x * wic.x + wic.y * wic.y) * (P.ieta * P.ieta);
                if (s2o >= 1.0f) return invalid_sample();
                float c_out = sqrtf(1.0f - s2o);
                float T_out = 1.0f - fresnel_dielectric(c_out, P.ieta);
                S4 ltr = s4(1.0f);
                if (P.has_medium) {
                    float tr_in = layer_tr(P.thickness, mk3(0, 0, ct_in)), tr_out = layer_tr(P.thickness, mk3(0, 0, wic.z));
                    ltr = tr_in * tr_out * P.albedo;
                }
                wl = normalize(mk3(wic.x * P.ieta, wic.y * P.ieta, c_out));
                out.f = fc * T_in * T_out * ltr;
                out.pdf = (1.0f - Fi) * (tr_pdf(woc, wm, cax, cay) / (4.0f * fabsf(com)));
            }
        }
    } else {
        float iax = maxf(P.iax, 1e-4f), iay = maxf(P.iay, 1e-4f);
        v3 wm = tr_sample_wm(wo, sample_u, iax, iay);
        float com = dot(wo, wm);
        if (com < 0.0f) return invalid_sample();
        float Fi = fresnel_dielectric(com, P.ieta);
        if (rng < Fi) {
            wl = -wo + 2.0f * com * wm;
            if (wl.z * wo.z < 0.0f) return invalid_sample();
            float D = tr_d(wm, iax, iay), G = tr_g(wo, wl, iax, iay);
            out.pdf = Fi * tr_pdf(wo, wm, iax, iay) / (4.0f * fabsf(com));
            out.f = s4(D * G / (4.0f * fabsf(wl.z) * fabsf(wo.z)));
        } else {
            float T_in = 1.0f - Fi;
            if (c_smooth) {
                wl = mk3(-wo.x, -wo.y, wo.z);
                float cb = fabsf(wl.z);
                S4 Fc = fr_complex4(cb, P.ce, P.ck);
                float T_out = 1.0f - fresnel_dielectric(cb, P.ieta);
                S4 ltr = s4(1.0f);
                if (P.has_medium) {
                    float tr = layer_tr(P.thickness, wl);
                    ltr = tr * tr * P.albedo;
                }
                out.f = Fc * T_in * T_out * ltr / cos_o;
                out.pdf = (1.0f - Fi) * tr_pdf(wo, wm, iax, iay) / (4.0f * fabsf(com));
            } else {
                float cax = maxf(P.cax, 1e-4f), cay = maxf(P.cay, 1e-4f);
                v3 wmc = tr_sample_wm(wo, sample_u, cax, cay);
                float comc = dot(wo, wmc);
                if (comc < 0.0f) return invalid_sample();
                wl = -wo + 2.0f * comc * wmc;
                if (wl.z * wo.z < 0.0f) return invalid_sample();
                S4 Fc = fr_complex4(fabsf(comc), P.ce, P.ck);
                float D = tr_d(wmc, cax, cay), G = tr_g(wo, wl, cax, cay);
                float ci = fabsf(wl.z), co = fabsf(wo.z);
                S4 fc = D * Fc * G / (4.0f * ci * co);
                float T_out = 1.0f - fresnel_dielectric(ci, P.ieta);
                S4 ltr = s4(1.0f);
                if (P.has_medium) {
                    float tr_in = layer_tr(P.thickness, mk3(0, 0, co)), tr_out = layer_tr(P.thickness, wl);
                    ltr = tr_in * tr_out * P.albedo;
                }
                out.f = fc * T_in * T_out * ltr;
                out.pdf = (1.0f - Fi) * tr_pdf(wo, wmc, cax, cay) / (4.0f * fabsf(comc));
            }
        }
    }
    if (flip) wl = -wl;
    out.wi = normalize(t * wl.x + b * wl.y + n * wl.z);
    return out;
}
HKD S4 cc_eval(const CCParams& P, v3 wo_w, v3 wi_w, v3 n, float& pdf) {
    pdf = 0.0f;
    float ci = dot(wi_w, n), co = dot(wo_w, n);
    if (ci * co < 0.0f) return s4(0.0f);
    if (fabsf(ci) < 1e-6f || fabsf(co) < 1e-6f) return s4(0.0f);
    v3 t, b;
    coordinate_system(n, t, b);
    v3 wo = mk3(dot(wo_w, t), dot(wo_w, b), co);
    v3 wi = mk3(dot(wi_w, t), dot(wi_w, b), ci);
    if (wo.z < 0.0f) {
        wo = -wo;
        wi = -wi;
    }
    bool i_smooth = tr_smooth(P.iax, P.iay), c_smooth = tr_smooth(P.cax, P.cay);
    if (i_smooth && c_smooth) return s4(0.0f);
    v3 wh = normalize(wo + wi);
    if (wh.z < 0.0f) wh = -wh;
    float coh = dot(wo, wh);
    float T_o = 1.0f - fresnel_dielectric(fabsf(wo.z), P.ieta);
    float T_i = 1.0f - fresnel_dielectric(fabsf(wi.z), P.ieta);
    S4 ltr = s4(1.0f);
    if (P.has_medium) {
        float tr = layer_tr(P.thickness, wi);
        ltr = tr * tr * P.albedo;
    }
    if (i_smooth) {
        float cax = maxf(P.cax, 1e-4f), cay = maxf(P.cay, 1e-4f);
        float D = tr_d(wh, cax, cay), G = tr_g(wo, wi, cax, cay);
        S4 Fc = fr_complex4(fabsf(coh), P.ce, P.ck);
        S4 fc = D * Fc * G / (4.0f * fabsf(wi.z) * fabsf(wo.z));
        pdf = T_o * tr_pdf(wo, wh, cax, cay) / (4.0f * fabsf(coh));
        return fc * T_o * T_i * ltr;
    }
    float iax = maxf(P.iax, 1e-4f), iay = maxf(P.iay, 1e-4f);
    float F_wh = fresnel_dielectric(fabsf(coh), P.ieta);
    float F_o = fresnel_dielectric(fabsf(wo.z), P.ieta);
    float f_interface = tr_d(wh, iax, iay) * F_wh * tr_g(wo, wi, iax, iay) / (4.0f * fabsf(wi.z) * fabsf(wo.z));
    S4 fc;
    float pdf_c;
    if (c_smooth) {
        fc = fr_complex4(fabsf(wo.z), P.ce, P.ck) / fabsf(wo.z);
        pdf_c = 1.0f;
    } else {
        float cax = maxf(P.cax, 1e-4f), cay = maxf(P.cay, 1e-4f);
        S4 Fc = fr_complex4(fabsf(coh), P.ce, P.ck);
        fc = tr_d(wh, cax, cay) * Fc * tr_g(wo, wi, cax, cay) / (4.0f * fabsf(wi.z) * fabsf(wo.z));
        pdf_c = tr_pdf(wo, wh, cax, cay) / (4.0f * fabsf(coh));
    }
    S4 contrib = fc * T_o * T_i * ltr;
    float pdf_i = F_o * tr_pdf(wo, wh, iax, iay) / (4.0f * fabsf(coh));
    pdf = pdf_i + T_o * pdf_c;
    return s4(f_interface) + contrib;
}

// ---- ThinDielectric :1975-2051 ---------------------------------------------------------------------
HKD BSDFSample thin_dielectric_sample(float eta, v3 wo_w, v3 n, float rng) {
    float wdn = dot(wo_w, n);
    if (fabsf(wdn) < 1e-6f) return invalid_sample();
    v3 t, b;
    coordinate_system(n, t, b);
    v3 wo = mk3(dot(wo_w, t), dot(wo_w, b), wdn);
    float cos_o = fabsf(wo.z);
    float R0 = fresnel_dielectric(cos_o, eta), T0 = 1.0f - R0;
    float R = R0;
    if (R0 < 1.0f) R = R0 + T0 * T0 * R0 / (1.0f - R0 * R0);
    float Tt = 1.0f - R;
    if (R + Tt < 1e-10f) return invalid_sample();
    float prob_r = R / (R + Tt);
    BSDFSample s;
    s.is_specular = true, s.eta_scale = 1.0f;
    if (rng < prob_r) {
        v3 wl = mk3(-wo.x, -wo.y, wo.z);
        s.wi = normalize(t * wl.x + b * wl.y + n * wl.z);
        s.f = s4(R / fabsf(wl.z));
        s.pdf = prob_r;
        return s;
    }
    s.wi = -wo_w;
    s.f = s4(Tt / cos_o);
    s.pdf = 1.0f - prob_r;
    return s;
}

// ---- DiffuseTransmission :2083-2218 ----------------------------------------------------------------
struct DTParams {
    S4 r, t;
    float pr, pt;
};
HKD DTParams dt_params(const DScene& sc, const DTables& T, const DMaterial& m, const TexCtx& uv, S4 lambda) {
    // the host bakes clamp(rgb * scale, 0, 1) when both the colour and the scale are constants
    float scale = eval_f32(sc, m, 0, uv);
    bool baked = m.ftex[0] < 0;
    DTParams P;
    float raw[4];
    rgb_param_raw(sc, m.rgb[0], uv, raw);
    float r0 = clampf(raw[0] * scale, 0.0f, 1.0f), r1 = clampf(raw[1] * scale, 0.0f, 1.0f), r2 = clampf(raw[2] * scale, 0.0f, 1.0f);
    P.r = eval_bounded((baked && m.rgb[0].tex < 0) ? m.rgb[0].coef : coef_bounded(T, r0, r1, r2), lambda);
    P.pr = maxf(maxf(r0, r1), r2);
    rgb_param_raw(sc, m.rgb[1], uv, raw);
    r0 = clampf(raw[0] * scale, 0.0f, 1.0f), r1 = clampf(raw[1] * scale, 0.0f, 1.0f), r2 = clampf(raw[2] * scale, 0.0f, 1.0f);
    P.t = eval_bounded((baked && m.rgb[1].tex < 0) ? m.rgb[1].coef : coef_bounded(T, r0, r1, r2), lambda);
    P.pt = maxf(maxf(r0, r1), r2);
    return P;
}
HKD BSDFSample dt_sample(const DTParams& P, v3 wo_w, v3 n, v2 u, float rng) {
    float wdn = dot(wo_w, n);
    if (fabsf(wdn) < 1e-6f) return invalid_sample();
    if (P.pr + P.pt < 1e-10f) return invalid_sample();
    v3 t, b;
    coordinate_system(n, t, b);
    float prob_r = P.pr / (P.pr + P.pt);
    bool refl = rng < prob_r;
    v3 lw = cosine_sample_hemisphere(u);
    if (refl ? (wdn < 0.0f) : (wdn > 0.0f)) lw = mk3(lw.x, lw.y, -lw.z);
    float ct = fabsf(lw.z);
    if (ct < 1e-6f) return invalid_sample();
    BSDFSample s;
    s.wi = normalize(t * lw.x + b * lw.y + n * lw.z);
    s.f = (refl ? P.r : P.t) * (1.0f / PI_F);
    s.pdf = (refl ? prob_r : (1.0f - prob_r)) * ct / PI_F;
    s.is_specular = false, s.eta_scale = 1.0f;
    return s;
}
HKD S4 dt_eval(const DTParams& P, v3 wo_w, v3 wi_w, v3 n, float& pdf) {
    pdf = 0.0f;
    float ci = dot(wi_w, n), co = dot(wo_w, n);
    float aci = fabsf(ci);
    if (aci < 1e-6f) return s4(0.0f);
    if (P.pr + P.pt < 1e-10f) return s4(0.0f);
    if (ci * co > 0.0f) {
        pdf = (P.pr / (P.pr + P.pt)) * aci / PI_F;
        return P.r * (1.0f / PI_F);
    }
    pdf = (P.pt / (P.pr + P.pt)) * aci / PI_F;
    return P.t * (1.0f / PI_F);
}

}  // namespace hkd

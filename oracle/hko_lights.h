// hko_lights.h — CPU ORACLE (test infrastructure): light sampling and the light BVH.
// Follows:
//   sample_light_spectral (per type)       src/integrators/physical-wavefront/lights.jl:39-297
//   compute_direct_lighting_spectral       src/integrators/physical-wavefront/lights.jl:535-600
//   arealight_Le                           src/lights/diffuse-area.jl:53-82
//   DirectionCone / LightBounds / union    src/lights/light-bounds.jl:24-158
//   light_bounds(light)                    src/lights/light-bounds.jl:231-295
//   BVHLightSampler build                  src/lights/bvh-light-sampler.jl:242-466
//   node_importance / sample / pmf         src/lights/bvh-light-sampler.jl:58-232
#pragma once
#include <vector>

#include "hikari_mi355x.h"
#include "hko_accel.h"
#include "hko_spectral.h"

namespace hko {

struct TextureSet {
    const hk_texture* tex = nullptr;
    int32_t n = 0;
    const hk_envmap* envmaps = nullptr;  // EnvironmentLight.env_map records (the reference derefs them through `lights`)
};
// _sample_texture_bilinear  src/textures/texture-ref.jl:151-186
inline void tex_fetch(const hk_texture& t, int32_t y1b, int32_t x1b, float* out) {
    const float* p = t.data + ((size_t)(y1b - 1) + (size_t)t.height * (size_t)(x1b - 1)) * t.channels;
    for (int c = 0; c < t.channels; ++c) out[c] = p[c];
}
inline void sample_texture_bilinear(const hk_texture& t, V2 uv, float* out) {
    float ua0 = 1.0f - uv.y, ua1 = uv.x;
    int32_t h = t.height, w = t.width;
    float px = ua1 * (float)(w - 1) + 1.0f;
    float py = ua0 * (float)(h - 1) + 1.0f;
    int32_t x0 = (int32_t)std::floor(px), y0 = (int32_t)std::floor(py);
    int32_t x1 = x0 + 1, y1 = y0 + 1;
    x0 = clampi(x0, 1, w);
    x1 = clampi(x1, 1, w);
    y0 = clampi(y0, 1, h);
    y1 = clampi(y1, 1, h);
    float fx = px - std::floor(px), fy = py - std::floor(py);
    float c00[4], c10[4], c01[4], c11[4];
    tex_fetch(t, y0, x0, c00);
    tex_fetch(t, y0, x1, c10);
    tex_fetch(t, y1, x0, c01);
    tex_fetch(t, y1, x1, c11);
    for (int c = 0; c < t.channels; ++c) {
        float c0 = c00[c] * (1.0f - fx) + c10[c] * fx;
        float c1 = c01[c] * (1.0f - fx) + c11[c] * fx;
        out[c] = c0 * (1.0f - fy) + c1 * fy;
    }
}
// _sample_texture_data (textures/basic.jl:19-26): NEAREST texel by truncation, same (1-v, u) flip.  This is what
// eval_tex(ctx, ref, uv::Point2f) resolves to, i.e. the alpha test (spectral-eval.jl:3882-3885); shading goes through the
// TextureFilterContext method, which is bilinear (texture-ref.jl:71-74, 151-186).
inline void sample_texture_nearest(const hk_texture& t, V2 uv, float* out) {
    if (t.kind == 1) {  // eval_tex(::VertexColorTexture, ::Point2f) = RGBSpectrum(0.5f0)
        for (int c = 0; c < t.channels; ++c) out[c] = c < 3 ? 0.5f : 1.0f;
        return;
    }
    float a0 = 1.0f - uv.y, a1 = uv.x;
    int32_t i = clampi((int32_t)(1.0f + (float)(t.height - 1) * a0), 1, t.height);
    int32_t j = clampi((int32_t)(1.0f + (float)(t.width - 1) * a1), 1, t.width);
    tex_fetch(t, i, j, out);
}
inline RGBA eval_tex_nearest(const TextureSet& ts, const hk_tex_rgba& f, V2 uv) {
    if (f.tex < 0) return RGBA(f.c[0], f.c[1], f.c[2], f.c[3]);
    float o[4] = {0, 0, 0, 1};
    sample_texture_nearest(ts.tex[f.tex], uv, o);
    return RGBA(o[0], o[1], o[2], o[3]);
}
inline float eval_tex_nearest(const TextureSet& ts, const hk_tex_f32& f, V2 uv) {
    if (f.tex < 0) return f.v;
    float o[4] = {0, 0, 0, 0};
    sample_texture_nearest(ts.tex[f.tex], uv, o);
    return o[0];
}
// TextureFilterContext as far as the path reads it (texture-ref.jl:21-29): uv + (face_idx, bary) for vertex colours.
// A bare uv converts implicitly (face 0: a VertexColorTexture then yields its gray placeholder, texture-ref.jl:245).
struct TexCtx {
    V2 uv;
    uint32_t face = 0;
    float bary[3] = {0, 0, 0};  // (w, u, v)
    TexCtx(V2 u) : uv(u) {}
    TexCtx(V2 u, uint32_t f, const float b[3]) : uv(u), face(f), bary{b[0], b[1], b[2]} {}
};
// eval_tex(ctx, tex, tfc): bilinear for images (texture-ref.jl:71-74), barycentric for VertexColorTexture (:230-235)
inline void sample_texture_ctx(const hk_texture& t, const TexCtx& tc, float* out) {
    if (t.kind == 1) {
        if (tc.face == 0) {
            for (int c = 0; c < t.channels; ++c) out[c] = c < 3 ? 0.5f : 1.0f;
            return;
        }
        const float* f = t.data + (size_t)(tc.face - 1) * 3 * (size_t)t.channels;
        for (int c = 0; c < t.channels; ++c) out[c] = f[c] * tc.bary[0] + f[t.channels + c] * tc.bary[1] + f[2 * t.channels + c] * tc.bary[2];
        return;
    }
    sample_texture_bilinear(t, tc.uv, out);
}
inline RGBA eval_tex(const TextureSet& ts, const hk_tex_rgba& f, const TexCtx& uv) {
    if (f.tex < 0) return RGBA(f.c[0], f.c[1], f.c[2], f.c[3]);
    float o[4] = {0, 0, 0, 1};
    sample_texture_ctx(ts.tex[f.tex], uv, o);
    return RGBA(o[0], o[1], o[2], o[3]);
}
inline float eval_tex(const TextureSet& ts, const hk_tex_f32& f, const TexCtx& uv) {
    if (f.tex < 0) return f.v;
    float o[4] = {0, 0, 0, 0};
    sample_texture_ctx(ts.tex[f.tex], uv, o);
    return o[0];
}

// ---- EnvironmentMap + Distribution2D -------------------------------------------------------------
//   equal_area_sphere_to_square / square_to_sphere   src/textures/environment_map.jl:78-160
//   direction_to_uv / uv_to_direction                src/textures/environment_map.jl:200-229
//   env(dir) bilinear :290-338 ; lookup_uv nearest :358-371
//   Distribution2D sample_continuous / pdf           src/sampler/sampling.jl:264-361
inline V2 equal_area_sphere_to_square(V3 d) {
    float x = std::fabs(d.x), y = std::fabs(d.y), z = std::fabs(d.z);
    // quirk Q35: sqrt(1 - |z|) of a direction whose |z| is 1 + 2^-23 after normalisation is undefined in the reference
    // (DomainError on the CPU, NaN on a GPU backend); max(0, .) is the identity wherever the reference is defined
    float r = std::sqrt(maxf(0.0f, 1.0f - z));
    float a = maxf(x, y);
    float b = a == 0.0f ? 0.0f : minf(x, y) / a;
    const float t1 = 0.406758566246788489601959989e-5f, t2 = 0.636226545274016134946890922156f, t3 = 0.61572017898280213493197203466e-2f,
                t4 = -0.247333733281268944196501420480f, t5 = 0.881770664775316294736387951347e-1f, t6 = 0.419038818029165735901852432784e-1f,
                t7 = -0.251390972343483509333252996350e-1f;
    float phi = t1 + b * (t2 + b * (t3 + b * (t4 + b * (t5 + b * (t6 + b * t7)))));
    if (x < y) phi = 1.0f - phi;
    float v = phi * r;
    float u = r - v;
    if (d.z < 0.0f) {
        float tu = u;
        u = v;
        v = tu;
        u = 1.0f - u;
        v = 1.0f - v;
    }
    u = std::copysign(u, d.x);
    v = std::copysign(v, d.y);
    return V2(0.5f * (u + 1.0f), 0.5f * (v + 1.0f));
}
inline V3 equal_area_square_to_sphere(V2 p) {
    float u = 2.0f * p.x - 1.0f, v = 2.0f * p.y - 1.0f;
    float up = std::fabs(u), vp = std::fabs(v);
    float sd = 1.0f - (up + vp);
    float d = std::fabs(sd);
    float r = 1.0f - d;
    float phi = (r == 0.0f ? 1.0f : (vp - up) / r + 1.0f) * PI_F / 4.0f;
    float z = std::copysign(1.0f - r * r, sd);
    float sphi, cphi;
    jl_sincos(phi, sphi, cphi);
    float cos_phi = std::copysign(cphi, u);
    float sin_phi = std::copysign(sphi, v);
    float r_cyl = r * std::sqrt(2.0f - r * r);
    return V3(cos_phi * r_cyl, sin_phi * r_cyl, z);
}
// rotation is the reference's Mat3f, row-major in the record: R[i][j] = rotation[3*i + j]
inline V3 env_rotate(const hk_envmap& e, V3 d) {  // rotation * d
    const float* R = e.rotation;
    return V3(R[0] * d.x + R[1] * d.y + R[2] * d.z, R[3] * d.x + R[4] * d.y + R[5] * d.z, R[6] * d.x + R[7] * d.y + R[8] * d.z);
}
inline V3 env_rotate_t(const hk_envmap& e, V3 d) {  // transpose(rotation) * d
    const float* R = e.rotation;
    return V3(R[0] * d.x + R[3] * d.y + R[6] * d.z, R[1] * d.x + R[4] * d.y + R[7] * d.z, R[2] * d.x + R[5] * d.y + R[8] * d.z);
}
inline V2 env_direction_to_uv(const hk_envmap& e, V3 dir) { return equal_area_sphere_to_square(env_rotate_t(e, dir)); }
inline V3 env_uv_to_direction(const hk_envmap& e, V2 uv) { return env_rotate(e, equal_area_square_to_sphere(uv)); }
inline RGBA env_texel(const hk_envmap& e, int32_t y1b, int32_t x1b) {  // data[y, x], Julia Matrix{RGBSpectrum}[h, w]
    const float* p = e.data + ((size_t)(y1b - 1) + (size_t)e.height * (size_t)(x1b - 1)) * 4;
    return RGBA(p[0], p[1], p[2], p[3]);
}
inline RGBA env_eval(const hk_envmap& e, V3 dir) {  // bilinear, :290-338
    V2 uv = env_direction_to_uv(e, dir);
    int32_t h = e.height, w = e.width;
    float x = uv.x * (float)(w - 1) + 1.0f;
    float y = uv.y * (float)(h - 1) + 1.0f;
    int32_t x0 = floor_int32(x), y0 = floor_int32(y);
    int32_t x1 = x0 + 1, y1 = y0 + 1;
    x0 = clampi(x0, 1, w);
    x1 = clampi(x1, 1, w);
    y0 = clampi(y0, 1, h);
    y1 = clampi(y1, 1, h);
    x1 = x1 > w ? 1 : x1;
    float fx = x - (float)floor_int32(x), fy = y - (float)floor_int32(y);
    RGBA c00 = env_texel(e, y0, x0), c10 = env_texel(e, y0, x1), c01 = env_texel(e, y1, x0), c11 = env_texel(e, y1, x1);
    RGBA o;
    for (int c = 0; c < 4; ++c) {
        float c0 = c00.c[c] * (1.0f - fx) + c10.c[c] * fx;
        float c1 = c01.c[c] * (1.0f - fx) + c11.c[c] * fx;
        o.c[c] = c0 * (1.0f - fy) + c1 * fy;
    }
    return o;
}
inline RGBA env_lookup_uv(const hk_envmap& e, V2 uv) {  // nearest, :358-371
    int32_t ui = clampi(floor_int32(uv.x * (float)e.width) + 1, 1, e.width);
    int32_t vi = clampi(floor_int32(uv.y * (float)e.height) + 1, 1, e.height);
    return env_texel(e, vi, ui);
}
// branchless 20-step search (sampling.jl:305-333): last index with cdf[idx] <= u, 1-based over n entries of stride 1
inline int32_t find_interval_binary20(const float* cdf, int32_t n, float u) {
    int32_t lo = 1, hi = n;
    for (int k = 0; k < 20; ++k) {
        int32_t mid = (lo + hi + 1) / 2;
        bool c = cdf[mid - 1] <= u;
        lo = c ? mid : lo;
        hi = c ? hi : mid - 1;
    }
    return lo;
}
inline V2 dist2d_sample(const hk_envmap& e, V2 u, float& pdf) {
    const int32_t nu = e.nu, nv = e.nv;
    int32_t vo = clampi(find_interval_binary20(e.marginal_cdf, nv + 1, u.y), 1, nv);
    float du_v = u.y - e.marginal_cdf[vo - 1];
    float den_v = e.marginal_cdf[vo] - e.marginal_cdf[vo - 1];
    if (den_v > 0.0f) du_v /= den_v;
    float v_s = ((float)(vo - 1) + du_v) / (float)nv;
    float pdf_v = e.marginal_func_int > 0.0f ? e.marginal_func[vo - 1] / e.marginal_func_int : 0.0f;
    const float* ccdf = e.conditional_cdf + (size_t)(vo - 1) * (size_t)(nu + 1);
    int32_t uo = clampi(find_interval_binary20(ccdf, nu + 1, u.x), 1, nu);
    float du_u = u.x - ccdf[uo - 1];
    float den_u = ccdf[uo] - ccdf[uo - 1];
    if (den_u > 0.0f) du_u /= den_u;
    float u_s = ((float)(uo - 1) + du_u) / (float)nu;
    float fiv = e.conditional_func_int[vo - 1];
    float pdf_u = fiv > 0.0f ? e.conditional_func[(size_t)(vo - 1) * nu + (uo - 1)] / fiv : 0.0f;
    pdf = pdf_u * pdf_v;
    return V2(u_s, v_s);
}
inline float dist2d_pdf(const hk_envmap& e, V2 uv) {
    int32_t iu = clampi(floor_int32(uv.x * (float)e.nu) + 1, 1, e.nu);
    int32_t iv = clampi(floor_int32(uv.y * (float)e.nv) + 1, 1, e.nv);
    return e.conditional_func[(size_t)(iv - 1) * e.nu + (iu - 1)] / e.marginal_func_int;
}
inline RGBA rgba_mul(const RGBA& a, const RGBA& b) { return RGBA(a.c[0] * b.c[0], a.c[1] * b.c[1], a.c[2] * b.c[2], a.c[3] * b.c[3]); }
// pdf_li_spectral(EnvironmentLight)  physical-wavefront/lights.jl:336-347
inline float env_pdf_li(const hk_envmap& e, V3 wi) { return dist2d_pdf(e, env_direction_to_uv(e, wi)) / (4.0f * PI_F); }

struct LightSample {
    Spec Li;
    V3 wi = V3(0, 0, 1);
    float pdf = 0.0f;
    V3 p_light;
    bool is_delta = false;
};

// light.i evaluated as an illuminant: Sample(table, i, lambda)   uplift.jl:548-566
inline Spec light_spectrum(const RGB2SpecTable& t, const hk_light& l, const Wavelengths& w) {
    if (l.spectrum_kind == HK_SPEC_ILLUMINANT) return sample_illuminant(SigPoly{l.poly[0], l.poly[1], l.poly[2]}, l.illum_scale, w);
    return uplift_rgb_illuminant(t, RGBA(l.i_rgb[0], l.i_rgb[1], l.i_rgb[2], l.i_rgb[3]), w);
}

// arealight_Le   diffuse-area.jl:53-78 (bounded uplift: quirk Q3)
inline Spec arealight_Le(const RGB2SpecTable& t, const TextureSet& ts, const hk_light& l, V3 wo, V3 n, V2 uv, const Wavelengths& w) {
    if (l.kind != HK_LIGHT_DIFFUSE_AREA) return Spec();
    if (!l.two_sided && dot(wo, n) < 0.0f) return Spec();
    RGBA Le = eval_tex(ts, l.Le, uv) * l.scale;
    return uplift_rgb(t, Le, w);
}

inline LightSample sample_light_spectral(const RGB2SpecTable& t, const TextureSet& ts, const hk_light& l, V3 p, const Wavelengths& w, V2 u) {
    LightSample s;
    switch (l.kind) {
        case HK_LIGHT_POINT: {  // lights.jl:39-51
            V3 pos(l.position[0], l.position[1], l.position[2]);
            V3 to_light = pos - p;
            float dist_sq = dot(to_light, to_light);
            float dist = std::sqrt(dist_sq);
            if (dist < 1e-6f) return s;
            s.wi = to_light / dist;
            s.Li = (l.scale * light_spectrum(t, l, w)) / dist_sq;
            s.pdf = 1.0f;
            s.p_light = pos;
            s.is_delta = true;
            return s;
        }
        case HK_LIGHT_SPOT: {  // lights.jl:58-98
            V3 pos(l.position[0], l.position[1], l.position[2]);
            V3 to_light = pos - p;
            float dist_sq = dot(to_light, to_light);
            float dist = std::sqrt(dist_sq);
            if (dist < 1e-6f) return s;
            V3 wi = to_light / dist;
            const float* m = l.world_to_light;
            V3 mw = -wi;
            V3 wl = normalize(V3(m[0] * mw.x + m[1] * mw.y + m[2] * mw.z, m[4] * mw.x + m[5] * mw.y + m[6] * mw.z, m[8] * mw.x + m[9] * mw.y + m[10] * mw.z));
            float cos_theta = wl.z;
            if (cos_theta < l.cos_total_width) return s;
            float falloff;
            if (cos_theta >= l.cos_falloff_start)
                falloff = 1.0f;
            else {
                float delta = (cos_theta - l.cos_total_width) / (l.cos_falloff_start - l.cos_total_width);
                falloff = delta * delta * delta * delta;
            }
            s.wi = wi;
            s.Li = ((l.scale * light_spectrum(t, l, w)) * falloff) / dist_sq;
            s.pdf = 1.0f;
            s.p_light = pos;
            s.is_delta = true;
            return s;
        }
        case HK_LIGHT_DIRECTIONAL:
        case HK_LIGHT_SUN: {  // lights.jl:105-131
            V3 wi = -V3(l.direction[0], l.direction[1], l.direction[2]);
            s.wi = wi;
            s.p_light = p + 1.0e6f * wi;
            s.Li = l.scale * light_spectrum(t, l, w);
            s.pdf = 1.0f;
            s.is_delta = true;
            return s;
        }
        case HK_LIGHT_AMBIENT: {  // lights.jl:167-181
            float z = 1.0f - 2.0f * u.x;
            float r = std::sqrt(maxf(0.0f, 1.0f - z * z));
            float phi = 2.0f * PI_F * u.y;
            float sphi, cphi;
            jl_sincos(phi, sphi, cphi);
            V3 wi(r * cphi, r * sphi, z);
            s.wi = wi;
            s.pdf = 1.0f / (4.0f * PI_F);
            s.p_light = p + 1.0e6f * wi;
            s.Li = l.scale * light_spectrum(t, l, w);
            s.is_delta = false;
            return s;
        }
        case HK_LIGHT_ENVIRONMENT: {  // lights.jl:158-190
            const hk_envmap& e = ts.envmaps[l.envmap];
            float map_pdf;
            V2 uv = dist2d_sample(e, u, map_pdf);
            V3 wi = env_uv_to_direction(e, uv);
            float pdf = map_pdf / (4.0f * PI_F);
            if (pdf <= 0.0f) return s;
            RGBA Li_rgb = rgba_mul(env_lookup_uv(e, uv), RGBA(l.i_rgb[0], l.i_rgb[1], l.i_rgb[2], l.i_rgb[3]));
            s.wi = wi;
            s.pdf = pdf;
            s.p_light = p + 1.0e6f * wi;
            s.Li = uplift_rgb_illuminant(t, Li_rgb, w);
            s.is_delta = false;
            return s;
        }
        case HK_LIGHT_DIFFUSE_AREA: {  // lights.jl:190-247
            float b0, b1;
            if (u.x < u.y) {
                b0 = u.x / 2.0f;
                b1 = u.y - b0;
            } else {
                b1 = u.y / 2.0f;
                b0 = u.x - b1;
            }
            float b2 = 1.0f - b0 - b1;
            V3 v0(l.v[0], l.v[1], l.v[2]), v1(l.v[3], l.v[4], l.v[5]), v2(l.v[6], l.v[7], l.v[8]);
            V3 p_light = b0 * v0 + b1 * v1 + b2 * v2;
            V3 to_light = p_light - p;
            float dist_sq = dot(to_light, to_light);
            if (dist_sq < 1e-12f) return s;
            float dist = std::sqrt(dist_sq);
            V3 wi = to_light / dist;
            V3 ln(l.normal[0], l.normal[1], l.normal[2]);
            float cos_theta = std::fabs(dot(ln, -wi));
            if (cos_theta < 1e-6f) return s;
            float pdf = dist_sq / (cos_theta * l.area);
            V2 uvs(b0 * l.uv[0] + b1 * l.uv[2] + b2 * l.uv[4], b0 * l.uv[1] + b1 * l.uv[3] + b2 * l.uv[5]);
            V3 wo(-wi.x, -wi.y, -wi.z);
            Spec Le = arealight_Le(t, ts, l, wo, ln, uvs, w);
            if (is_black(Le)) return s;
            s.Li = Le;
            s.wi = wi;
            s.pdf = pdf;
            s.p_light = p_light;
            s.is_delta = false;
            return s;
        }
        default: return s;  // environment: added with the env-map widening
    }
}

// ---- direction cones / light bounds ------------------------------------------------------------
struct DirectionCone {
    V3 w = V3(0, 0, 1);
    float cos_t = INF_F;
};
inline bool cone_empty(const DirectionCone& c) { return c.cos_t == INF_F; }
inline DirectionCone entire_sphere() { return DirectionCone{V3(0, 0, 1), -1.0f}; }
inline float angle_between(V3 a, V3 b) {
    if (dot(a, b) < 0.0f) return PI_F - 2.0f * std::asin(clampf(norm(a + b) * 0.5f, -1.0f, 1.0f));
    return 2.0f * std::asin(clampf(norm(b - a) * 0.5f, -1.0f, 1.0f));
}
inline DirectionCone cone_union(const DirectionCone& a, const DirectionCone& b) {
    if (cone_empty(a)) return b;
    if (cone_empty(b)) return a;
    float ta = std::acos(clampf(a.cos_t, -1.0f, 1.0f));
    float tb = std::acos(clampf(b.cos_t, -1.0f, 1.0f));
    float td = angle_between(a.w, b.w);
    if (minf(td + tb, PI_F) <= ta) return a;
    if (minf(td + ta, PI_F) <= tb) return b;
    float to = (ta + td + tb) * 0.5f;
    if (to >= PI_F) return entire_sphere();
    float tr = to - ta;
    V3 wr = cross(a.w, b.w);
    float len_sq = dot(wr, wr);
    if (len_sq == 0.0f) return entire_sphere();
    V3 axis = normalize(wr);
    float s = std::sin(tr), c = std::cos(tr);
    V3 w = a.w * c + cross(axis, a.w) * s + axis * dot(axis, a.w) * (1.0f - c);
    return DirectionCone{normalize(w), std::cos(to)};
}
struct Bounds3 {
    V3 lo = V3(INF_F), hi = V3(-INF_F);
};
inline Bounds3 bounds_union(const Bounds3& a, const Bounds3& b) {
    Bounds3 r;
    r.lo = V3(minf(a.lo.x, b.lo.x), minf(a.lo.y, b.lo.y), minf(a.lo.z, b.lo.z));
    r.hi = V3(maxf(a.hi.x, b.hi.x), maxf(a.hi.y, b.hi.y), maxf(a.hi.z, b.hi.z));
    return r;
}
inline float distance_squared(V3 a, V3 b) {
    V3 d = a - b;
    return dot(d, d);
}
struct LightBounds {
    Bounds3 bounds;
    V3 w = V3(0, 0, 1);
    float phi = 0.0f, cos_o = 1.0f, cos_e = 1.0f;
    bool two_sided = false;
};
inline V3 lb_centroid(const LightBounds& lb) { return (lb.bounds.lo + lb.bounds.hi) * 0.5f; }
inline LightBounds lb_union(const LightBounds& a, const LightBounds& b) {
    if (a.phi == 0.0f) return b;
    if (b.phi == 0.0f) return a;
    DirectionCone cone = cone_union(DirectionCone{a.w, a.cos_o}, DirectionCone{b.w, b.cos_o});
    LightBounds r;
    r.bounds = bounds_union(a.bounds, b.bounds);
    r.w = cone.w;
    r.phi = a.phi + b.phi;
    r.cos_o = cone.cos_t;
    r.cos_e = minf(a.cos_e, b.cos_e);
    r.two_sided = a.two_sided | b.two_sided;
    return r;
}
inline float cos_sub_clamped(float sinA, float cosA, float sinB, float cosB) { return cosA > cosB ? 1.0f : cosA * cosB + sinA * sinB; }
inline float sin_sub_clamped(float sinA, float cosA, float sinB, float cosB) { return cosA > cosB ? 0.0f : sinA * cosB - cosA * sinB; }

// luminance(light.i)   light-sampler.jl:448-456, rgb2spec.jl:353-355 (D65_MAX_VALUE = 100)
inline float light_i_luminance(const hk_light& l) {
    if (l.spectrum_kind == HK_SPEC_ILLUMINANT) return l.illum_scale * poly_max_value(SigPoly{l.poly[0], l.poly[1], l.poly[2]}) * 100.0f;
    return 0.212671f * l.i_rgb[0] + 0.715160f * l.i_rgb[1] + 0.072169f * l.i_rgb[2];
}
// light_bounds(light): returns false for infinite lights   light-bounds.jl:231-295
inline bool light_bounds(const hk_light& l, LightBounds& out) {
    const float cos_pi = (float)std::cos(3.14159265358979323846);       // Float32(cos(pi)) = -1
    const float cos_half_pi = (float)std::cos(3.14159265358979323846 / 2);  // Float32(cos(pi/2)) = 6.123234e-17
    switch (l.kind) {
        case HK_LIGHT_POINT: {
            V3 p(l.position[0], l.position[1], l.position[2]);
            out.bounds.lo = p;
            out.bounds.hi = p;
            out.w = V3(0, 0, 1);
            out.phi = 4.0f * PI_F * l.scale * light_i_luminance(l);
            out.cos_o = cos_pi;
            out.cos_e = cos_half_pi;
            out.two_sided = false;
            return true;
        }
        case HK_LIGHT_SPOT: {
            V3 p(l.position[0], l.position[1], l.position[2]);
            const float* m = l.light_to_world;
            V3 w = normalize(V3(m[2], m[6], m[10]));  // light_to_world(Vec3f(0,0,1))
            float cos_e = (float)std::cos(std::acos(l.cos_total_width) - std::acos(l.cos_falloff_start));
            if (cos_e == 1.0f && l.cos_total_width != l.cos_falloff_start) cos_e = 0.999f;
            out.bounds.lo = p;
            out.bounds.hi = p;
            out.w = w;
            out.phi = 4.0f * PI_F * l.scale * light_i_luminance(l);
            out.cos_o = l.cos_falloff_start;
            out.cos_e = cos_e;
            out.two_sided = false;
            return true;
        }
        case HK_LIGHT_DIFFUSE_AREA: {
            Bounds3 b;
            for (int k = 0; k < 3; ++k) {
                V3 v(l.v[3 * k], l.v[3 * k + 1], l.v[3 * k + 2]);
                Bounds3 pb;
                pb.lo = v;
                pb.hi = v;
                b = k == 0 ? pb : bounds_union(b, pb);
            }
            float sided = l.two_sided ? 2.0f : 1.0f;
            // Le isa RGBSpectrum ? luminance(Le) : scale   (textured: scale as proxy, quirk Q15)
            float Le_lum = l.Le.tex < 0 ? (0.212671f * l.Le.c[0] + 0.715160f * l.Le.c[1] + 0.072169f * l.Le.c[2]) : l.scale;
            out.bounds = b;
            out.w = V3(l.normal[0], l.normal[1], l.normal[2]);
            out.phi = PI_F * sided * l.area * l.scale * Le_lum;
            out.cos_o = 1.0f;
            out.cos_e = cos_half_pi;
            out.two_sided = l.two_sided != 0;
            return true;
        }
        default: return false;
    }
}

struct LightBVHNode {
    V3 bmin, bmax, w;
    float phi, cos_o, cos_e;
    bool two_sided;
    uint32_t child1_or_light;  // 1-based
    bool is_leaf;
};
inline LightBVHNode make_node(const LightBounds& lb, uint32_t c, bool leaf) {
    return LightBVHNode{lb.bounds.lo, lb.bounds.hi, lb.w, lb.phi, lb.cos_o, lb.cos_e, lb.two_sided, c, leaf};
}

// node_importance   bvh-light-sampler.jl:58-91 (d2 floor uses the diagonal *length*: quirk Q15)
inline float node_importance(const LightBVHNode& nd, V3 p, V3 n) {
    if (nd.phi == 0.0f) return 0.0f;
    V3 pc = (nd.bmin + nd.bmax) * 0.5f;
    float d2 = distance_squared(p, pc);
    d2 = maxf(d2, norm(nd.bmax - nd.bmin) * 0.5f);
    V3 wi = normalize(p - pc);
    float cos_w = dot(nd.w, wi);
    if (nd.two_sided) cos_w = std::fabs(cos_w);
    float sin_w = std::sqrt(maxf(0.0f, 1.0f - cos_w * cos_w));
    // bound_subtended_directions(bounds, p).cos   light-bounds.jl:96-109
    float cos_b;
    {
        V3 pcen = (nd.bmin + nd.bmax) * 0.5f;
        float radius_sq = distance_squared(nd.bmax, pcen);
        float dd = distance_squared(p, pcen);
        if (dd < radius_sq)
            cos_b = -1.0f;
        else {
            float sin2 = radius_sq / dd;
            cos_b = std::sqrt(maxf(0.0f, 1.0f - sin2));
        }
    }
    float sin_b = std::sqrt(maxf(0.0f, 1.0f - cos_b * cos_b));
    float sin_o = std::sqrt(maxf(0.0f, 1.0f - nd.cos_o * nd.cos_o));
    float cos_x = cos_sub_clamped(sin_w, cos_w, sin_o, nd.cos_o);
    float sin_x = sin_sub_clamped(sin_w, cos_w, sin_o, nd.cos_o);
    float cos_p = cos_sub_clamped(sin_x, cos_x, sin_b, cos_b);
    if (cos_p <= nd.cos_e) return 0.0f;
    float imp = nd.phi * cos_p / d2;
    if (n != V3(0.0f)) {
        float cos_i = std::fabs(dot(wi, n));
        float sin_i = std::sqrt(maxf(0.0f, 1.0f - cos_i * cos_i));
        float cos_pi = cos_sub_clamped(sin_i, cos_i, sin_b, cos_b);
        imp *= cos_pi;
    }
    return maxf(imp, 0.0f);
}

struct LightSampler {
    std::vector<LightBVHNode> nodes;
    std::vector<uint32_t> bit_trail;   // per light (0-based storage of 1-based flat index)
    std::vector<int32_t> infinite;     // 1-based flat indices
    int32_t num_bvh = 0, num_infinite = 0;
    mutable uint64_t nodes_evaluated = 0;

    // _evaluate_cost   bvh-light-sampler.jl:242-259
    static float evaluate_cost(const LightBounds& lb, const Bounds3& bounds, int dim) {
        float to = std::acos(clampf(lb.cos_o, -1.0f, 1.0f));
        float te = std::acos(clampf(lb.cos_e, -1.0f, 1.0f));
        float tw = minf(to + te, PI_F);
        float sin_o = std::sqrt(maxf(0.0f, 1.0f - lb.cos_o * lb.cos_o));
        float M = 2.0f * PI_F * (1.0f - lb.cos_o) + PI_F / 2.0f * (2.0f * tw * sin_o - std::cos(to - 2.0f * tw) - 2.0f * to * sin_o + lb.cos_o);
        V3 d = bounds.hi - bounds.lo;
        float max_d = maxf(maxf(d.x, d.y), d.z);
        float dim_d = d[dim];
        float Kr = dim_d > 1e-10f ? max_d / dim_d : max_d / 1e-10f;
        // surface_area(bounds) = 2(dx*dy + dx*dz + dy*dz)  (Raycore.surface_area, pbrt Bounds3::SurfaceArea)
        float sa = 2.0f * (d.x * d.y + d.x * d.z + d.y * d.z);
        return lb.phi * M * Kr * sa;
    }
    static int bucket_of(const Bounds3& cb, V3 c, int dim) {
        // Raycore.offset(bounds, p)[dim] = (p - pmin)/(pmax - pmin) when pmax > pmin (pbrt Bounds3::Offset)
        float o = c[dim] - cb.lo[dim];
        if (cb.hi[dim] > cb.lo[dim]) o /= (cb.hi[dim] - cb.lo[dim]);
        int b = (int)std::floor(12 * o);
        if (b < 0) b = 0;
        if (b > 11) b = 11;
        return b + 1;
    }
    typedef std::pair<int32_t, LightBounds> Item;
    LightBounds build_rec(std::vector<Item>& L, int start, int stop, uint32_t trail, int depth) {
        int count = stop - start + 1;
        if (count == 1) {
            nodes.push_back(make_node(L[start - 1].second, (uint32_t)L[start - 1].first, true));
            bit_trail[L[start - 1].first - 1] = trail;
            return L[start - 1].second;
        }
        LightBounds overall = L[start - 1].second;
        Bounds3 cb;
        {
            V3 c = lb_centroid(L[start - 1].second);
            cb.lo = c;
            cb.hi = c;
        }
        for (int i = start + 1; i <= stop; ++i) {
            overall = lb_union(overall, L[i - 1].second);
            V3 c = lb_centroid(L[i - 1].second);
            Bounds3 pb;
            pb.lo = c;
            pb.hi = c;
            cb = bounds_union(cb, pb);
        }
        float best_cost = INF_F;
        int best_dim = 0, best_bucket = 0;
        for (int dim = 1; dim <= 3; ++dim) {
            float extent = cb.hi[dim - 1] - cb.lo[dim - 1];
            if (extent <= 0.0f) continue;
            LightBounds bb[12];
            int bc[12] = {0};
            for (int i = start; i <= stop; ++i) {
                int b = bucket_of(cb, lb_centroid(L[i - 1].second), dim - 1);
                bb[b - 1] = lb_union(bb[b - 1], L[i - 1].second);
                bc[b - 1] += 1;
            }
            for (int split = 1; split <= 11; ++split) {
                LightBounds below, above;
                int cbelow = 0, cabove = 0;
                for (int b = 1; b <= split; ++b) {
                    below = lb_union(below, bb[b - 1]);
                    cbelow += bc[b - 1];
                }
                for (int b = split + 1; b <= 12; ++b) {
                    above = lb_union(above, bb[b - 1]);
                    cabove += bc[b - 1];
                }
                if (cbelow == 0 || cabove == 0) continue;
                float cost = evaluate_cost(below, overall.bounds, dim - 1) + evaluate_cost(above, overall.bounds, dim - 1);
                if (cost < best_cost) {
                    best_cost = cost;
                    best_dim = dim;
                    best_bucket = split;
                }
            }
        }
        int mid;
        if (best_dim > 0) {
            int pivot = start;
            for (int i = start; i <= stop; ++i) {
                int b = bucket_of(cb, lb_centroid(L[i - 1].second), best_dim - 1);
                if (b <= best_bucket) {
                    if (i != pivot) std::swap(L[pivot - 1], L[i - 1]);
                    pivot += 1;
                }
            }
            if (pivot == start || pivot > stop)
                mid = start + count / 2;
            else
                mid = pivot - 1;
        } else {
            mid = start + count / 2 - 1;
        }
        mid = mid < start ? start : (mid > stop - 1 ? stop - 1 : mid);
        size_t node_idx = nodes.size();
        nodes.push_back(make_node(overall, 0, false));
        LightBounds lb0 = build_rec(L, start, mid, trail, depth + 1);
        uint32_t child1 = (uint32_t)nodes.size() + 1;
        uint32_t bit = depth >= 32 ? 0u : (1u << depth);
        LightBounds lb1 = build_rec(L, mid + 1, stop, trail | bit, depth + 1);
        LightBounds merged = lb_union(lb0, lb1);
        nodes[node_idx] = make_node(merged, child1, false);
        return merged;
    }
    void build(const hk_light* lights, int32_t n) {
        nodes.clear();
        infinite.clear();
        bit_trail.assign(n, 0xFFFFFFFFu);
        std::vector<Item> L;
        for (int32_t i = 1; i <= n; ++i) {
            LightBounds lb;
            if (!light_bounds(lights[i - 1], lb))
                infinite.push_back(i);
            else if (lb.phi > 0.0f)
                L.push_back(Item(i, lb));
        }
        num_infinite = (int32_t)infinite.size();
        num_bvh = (int32_t)L.size();
        if (!L.empty()) build_rec(L, 1, (int)L.size(), 0u, 0);
    }

    // bvh_sample_light   bvh-light-sampler.jl:105-170 ; returns 1-based flat index (0 = failure)
    int32_t sample(V3 p, V3 n, float u, float& pmf_out) const {
        pmf_out = 0.0f;
        int32_t total = num_infinite + num_bvh;
        if (total == 0) return 0;
        bool has_bvh = num_bvh > 0;
        float p_inf = (float)num_infinite / (float)(num_infinite + (has_bvh ? 1 : 0));
        if (num_infinite > 0 && u < p_inf) {
            float ur = u / p_inf;
            int32_t idx = floor_int32(ur * (float)num_infinite);
            idx = (idx < num_infinite - 1 ? idx : num_infinite - 1) + 1;
            pmf_out = p_inf / (float)num_infinite;
            return infinite[idx - 1];
        }
        if (!has_bvh) return 0;
        float ub = num_infinite > 0 ? minf((u - p_inf) / (1.0f - p_inf), 0.99999994f) : minf(u, 0.99999994f);
        float pmf = 1.0f - p_inf;
        int32_t ni = 1;
        for (int it = 0; it < 64; ++it) {
            const LightBVHNode& nd = nodes[ni - 1];
            if (nd.is_leaf) {
                pmf_out = pmf;
                return (int32_t)nd.child1_or_light;
            }
            int32_t c0i = ni + 1, c1i = (int32_t)nd.child1_or_light;
            float c0 = node_importance(nodes[c0i - 1], p, n);
            float c1 = node_importance(nodes[c1i - 1], p, n);
            nodes_evaluated += 2;
            if (c0 == 0.0f && c1 == 0.0f) return 0;
            float sum = c0 + c1;
            float p0 = c0 / sum;
            if (ub < p0) {
                pmf *= p0;
                ub = ub / p0;
                ni = c0i;
            } else {
                pmf *= (1.0f - p0);
                ub = (ub - p0) / (1.0f - p0);
                ni = c1i;
            }
        }
        return 0;
    }
    // bvh_pmf   bvh-light-sampler.jl:184-232
    float pmf(V3 p, V3 n, int32_t light_1based) const {
        if (light_1based < 1) return 0.0f;
        bool has_bvh = num_bvh > 0;
        uint32_t trail = bit_trail[light_1based - 1];
        if (trail == 0xFFFFFFFFu) {
            if (num_infinite == 0) return 0.0f;
            return 1.0f / (float)(num_infinite + (has_bvh ? 1 : 0));
        }
        if (!has_bvh) return 0.0f;
        float p_inf = (float)num_infinite / (float)(num_infinite + 1);
        float pm = 1.0f - p_inf;
        int32_t ni = 1;
        for (int it = 0; it < 64; ++it) {
            const LightBVHNode& nd = nodes[ni - 1];
            if (nd.is_leaf) return pm;
            int32_t c0i = ni + 1, c1i = (int32_t)nd.child1_or_light;
            float c0 = node_importance(nodes[c0i - 1], p, n);
            float c1 = node_importance(nodes[c1i - 1], p, n);
            nodes_evaluated += 2;
            float sum = c0 + c1;
            if (sum <= 0.0f) return 0.0f;
            if ((trail & 1u) == 0u) {
                pm *= c0 / sum;
                ni = c0i;
            } else {
                pm *= c1 / sum;
                ni = c1i;
            }
            trail >>= 1;
        }
        return pm;
    }
};

}  // namespace hko

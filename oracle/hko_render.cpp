// hko_render.cpp — CPU ORACLE (test infrastructure; NOT part of the product, never a fallback).
//
// Restatement of Hikari.jl's VolPath wavefront path for checking the HIP implementation:
//   render! (one sample)                 src/integrators/volpath/volpath.jl:445-636
//   K1  vp_generate_camera_rays_kernel!  src/integrators/volpath/volpath.jl:125-205
//   K2  vp_generate_ray_samples_kernel!  src/integrators/volpath/volpath.jl:222-271
//   K3  vp_trace_rays_kernel!            src/integrators/volpath/intersection.jl:188-269
//   K7  vp_handle_escaped_rays_kernel!   src/integrators/volpath/intersection.jl:622-678
//   K8  vp_process_surface_hits_kernel!  src/integrators/volpath/surface-eval.jl:147-220
//   K9  surface_direct_lighting_inner!   src/integrators/volpath/surface-eval.jl:250-341
//   K10 vp_trace_shadow_rays_kernel!     src/integrators/volpath/intersection.jl:302-406, 565-600
//   K11 evaluate_material_inner!         src/integrators/volpath/surface-eval.jl:396-512
//   K12 vp_accumulate_to_rgb_kernel!     src/integrators/volpath/volpath.jl:326-375
//   K13 vp_finalize_film_kernel!         src/integrators/volpath/volpath.jl:384-417
// The stage order inside a bounce is the reference's; each stage is an OpenMP loop over its queue
// (one item per pixel per queue, so the order inside a stage cannot change any result).
//
// Parity status: "parity unpinned" at the Raycore boundary (see hko_accel.h) — the reference cannot
// be executed here (Julia absent) and its tests pin no radiance values; the pieces that
// have known answers (hashes, PCG32, Sobol, filter, Fresnel, uplift identities) are pinned by
// tests/test_oracle_kat.py.
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "hikari_mi355x.h"
#include "hko_layered.h"
#include "hko_filter_camera.h"
#include "hko_media.h"

#if defined(_OPENMP)
#include <omp.h>
#endif

namespace hko {

struct SetKeyM {  // medium index: -1 = vacuum (has_medium = valid)
    int32_t idx = -1;
};

struct RayItem {  // VPRayWorkItem  workitems.jl:14-41
    V3 o, d;
    float t_max, time;
    int32_t depth;
    Wavelengths lambda;
    int32_t pixel_index;  // 1-based
    Spec beta, r_u, r_l;
    float eta_scale;
    bool specular_bounce, any_non_specular;
    int32_t medium;
};
// Per-stage work arrays: entries are written before they are read (guarded by the stage's valid/kind flags), so they are left
// uninitialised — value-initialising ~1 KB per pixel per depth on one thread was the oracle's scaling limit on many-core hosts.
template <class T>
struct RawBuf {
    T* p;
    size_t n;
    explicit RawBuf(size_t count) : p((T*)std::malloc((count ? count : 1) * sizeof(T))), n(count) {}
    ~RawBuf() { std::free(p); }
    RawBuf(const RawBuf&) = delete;
    RawBuf& operator=(const RawBuf&) = delete;
    T& operator[](size_t i) { return p[i]; }
    const T& operator[](size_t i) const { return p[i]; }
    size_t size() const { return n; }
};
// growable array without value-initialisation (same reason), plus an order-preserving parallel compaction into it
template <class T>
struct RawVec {
    T* p = nullptr;
    size_t n = 0, cap = 0;
    RawVec() {}
    explicit RawVec(size_t count) { reserve(count), n = count; }
    ~RawVec() { std::free(p); }
    RawVec(const RawVec&) = delete;
    RawVec& operator=(const RawVec&) = delete;
    void reserve(size_t c) {
        if (c <= cap) return;
        p = (T*)std::realloc(p, c * sizeof(T));
        cap = c;
    }
    void push_back(const T& v) {
        if (n == cap) reserve(cap ? 2 * cap : 1024);
        std::memcpy(&p[n++], &v, sizeof(T));
    }
    T& operator[](size_t i) { return p[i]; }
    const T& operator[](size_t i) const { return p[i]; }
    size_t size() const { return n; }
    void swap(RawVec& o) {
        std::swap(p, o.p), std::swap(n, o.n), std::swap(cap, o.cap);
    }
    T* begin() { return p; }
    T* end() { return p + n; }
};
template <class T>
static void append_compact(RawVec<T>& dst, const T* src, const uint8_t* valid, int32_t n) {
#if defined(_OPENMP)
    const int nt = omp_get_max_threads();
#else
    const int nt = 1;
#endif
    std::vector<size_t> cnt((size_t)nt + 1, 0);
    const int32_t chunk = (n + nt - 1) / nt;
#pragma omp parallel for schedule(static, 1)
    for (int t = 0; t < nt; ++t) {
        int32_t lo = t * chunk, hi = lo + chunk < n ? lo + chunk : n;
        size_t c = 0;
        for (int32_t i = lo; i < hi; ++i) c += valid[i] ? 1 : 0;
        cnt[(size_t)t + 1] = c;
    }
    for (int t = 0; t < nt; ++t) cnt[(size_t)t + 1] += cnt[(size_t)t];
    const size_t base = dst.n;
    dst.reserve(base + cnt[(size_t)nt]);
#pragma omp parallel for schedule(static, 1)
    for (int t = 0; t < nt; ++t) {
        int32_t lo = t * chunk, hi = lo + chunk < n ? lo + chunk : n;
        size_t o = base + cnt[(size_t)t];
        for (int32_t i = lo; i < hi; ++i)
            if (valid[i]) std::memcpy(&dst.p[o++], &src[i], sizeof(T));
    }
    dst.n = base + cnt[(size_t)nt];
}
struct SurfaceGeom {
    V3 pi, n, dpdu, dpdv, ns, dpdus, dpdvs;
    V2 uv;
};
struct HitItem {  // VPHitSurfaceWorkItem
    V3 ray_o, ray_d;
    float ray_time;
    SurfaceGeom g;
    int32_t material;
    hk_medium_interface iface;
    uint32_t face_idx;
    float bary[3];
    uint32_t arealight;
    float triangle_area;
    Wavelengths lambda;
    int32_t pixel_index;
    Spec beta, r_u, r_l;
    int32_t depth;
    float eta_scale;
    bool specular_bounce, any_non_specular;
    int32_t current_medium;
    float t_hit;
};
struct MatItem {  // VPMaterialEvalWorkItem
    SurfaceGeom g;
    uint32_t face_idx;
    float bary[3];
    V3 wo;
    int32_t material;
    hk_medium_interface iface;
    Wavelengths lambda;
    int32_t pixel_index;
    Spec beta, r_u, r_l;
    int32_t depth;
    float eta_scale;
    bool specular_bounce, any_non_specular;
    int32_t current_medium;
};
struct ShadowItem {  // VPShadowRayWorkItem
    V3 o, d;
    float t_max;
    Wavelengths lambda;
    Spec Ld, r_u, r_l;
    int32_t pixel_index;
    int32_t medium;
};
struct EscapedItem {
    V3 d;
    Wavelengths lambda;
    int32_t pixel_index;
    Spec beta, r_u, r_l;
    int32_t depth;
    bool specular_bounce;
};

struct Scene {
    hk_scene_desc desc;
    Accel accel;
    LightSampler sampler;
    RGB2SpecTable table;
    CIETable cie;
    const uint32_t* sobol;
    MaterialCtx mctx;
    TextureSet textures;
    MediaCtx media;
    bool any_infinite_or_env = false;
};

struct Counters {
    uint64_t rays_closest = 0, rays_shadow = 0, nodes = 0, tris = 0, hits = 0, vertices = 0, collisions = 0;
};

// ---- geometry at a hit (intersection.jl:13-182) -------------------------------------------------
static inline void tri_vertices(const Scene& sc, int32_t prim, V3& v0, V3& v1, V3& v2) {
    const float* p = sc.desc.positions + 9 * (size_t)prim;
    v0 = V3(p[0], p[1], p[2]);
    v1 = V3(p[3], p[4], p[5]);
    v2 = V3(p[6], p[7], p[8]);
}
static inline V3 geometric_normal(const Scene& sc, int32_t prim) {
    V3 v0, v1, v2;
    tri_vertices(sc, prim, v0, v1, v2);
    return normalize(cross(v1 - v0, v2 - v0));
}
static inline void tri_uvs(const Scene& sc, int32_t prim, V2& a, V2& b, V2& c) {
    if (sc.desc.uvs) {
        const float* q = sc.desc.uvs + 6 * (size_t)prim;
        a = V2(q[0], q[1]);
        b = V2(q[2], q[3]);
        c = V2(q[4], q[5]);
    } else {
        a = V2(0, 0);
        b = V2(1, 0);
        c = V2(1, 1);
    }
}
static inline V2 uv_barycentric(const Scene& sc, int32_t prim, const float bary[3]) {
    V2 a, b, c;
    tri_uvs(sc, prim, a, b, c);
    float w = bary[0], u = bary[1], v = bary[2];
    return V2(w * a.x + u * b.x + v * c.x, w * a.y + u * b.y + v * c.y);
}
static inline float triangle_area(const Scene& sc, int32_t prim) {
    V3 v0, v1, v2;
    tri_vertices(sc, prim, v0, v1, v2);
    return 0.5f * norm(cross(v1 - v0, v2 - v0));
}
static SurfaceGeom surface_geometry(const Scene& sc, int32_t prim, const float bary[3], V3 ro, V3 rd, float t_hit) {
    SurfaceGeom g;
    g.pi = ro + rd * t_hit;
    V3 v0, v1, v2;
    tri_vertices(sc, prim, v0, v1, v2);
    V3 n = normalize(cross(v1 - v0, v2 - v0));
    g.uv = uv_barycentric(sc, prim, bary);
    // partial derivatives
    V2 uv0, uv1, uv2;
    tri_uvs(sc, prim, uv0, uv1, uv2);
    float du10 = uv1.x - uv0.x, dv10 = uv1.y - uv0.y, du20 = uv2.x - uv0.x, dv20 = uv2.y - uv0.y;
    V3 dp10 = v1 - v0, dp20 = v2 - v0;
    float det = du10 * dv20 - dv10 * du20;
    if (std::fabs(det) < 1e-8f) {
        V3 e1 = normalize(dp10);
        V3 nn = normalize(cross(dp10, dp20));
        g.dpdu = e1;
        g.dpdv = cross(nn, e1);
    } else {
        float inv_det = 1.0f / det;
        g.dpdu = (dv20 * dp10 - dv10 * dp20) * inv_det;
        g.dpdv = (-du20 * dp10 + du10 * dp20) * inv_det;
    }
    // shading normal
    V3 ns = n;
    bool has_normals = false;
    V3 n0, n1, n2;
    if (sc.desc.normals) {
        const float* q = sc.desc.normals + 9 * (size_t)prim;
        n0 = V3(q[0], q[1], q[2]);
        n1 = V3(q[3], q[4], q[5]);
        n2 = V3(q[6], q[7], q[8]);
        has_normals = !(std::isnan(n0.x) || std::isnan(n1.x) || std::isnan(n2.x));
    }
    float w = bary[0], u = bary[1], v = bary[2];
    if (has_normals) ns = normalize(V3(w * n0.x + u * n1.x + v * n2.x, w * n0.y + u * n1.y + v * n2.y, w * n0.z + u * n1.z + v * n2.z));
    n = dot(n, ns) < 0.0f ? -n : n;
    g.n = n;
    g.ns = ns;
    // shading tangents
    bool has_tangents = false;
    V3 t0, t1, t2;
    if (sc.desc.tangents) {
        const float* q = sc.desc.tangents + 9 * (size_t)prim;
        t0 = V3(q[0], q[1], q[2]);
        t1 = V3(q[3], q[4], q[5]);
        t2 = V3(q[6], q[7], q[8]);
        has_tangents = !std::isnan(t0.x) && !std::isnan(t1.x) && !std::isnan(t2.x);
    }
    V3 dpdus;
    if (has_tangents) {
        dpdus = normalize(V3(w * t0.x + u * t1.x + v * t2.x, w * t0.y + u * t1.y + v * t2.y, w * t0.z + u * t1.z + v * t2.z));
    } else {
        dpdus = g.dpdu - ns * dot(ns, g.dpdu);
        float len_sq = dot(dpdus, dpdus);
        if (len_sq > 1e-10f)
            dpdus = dpdus / std::sqrt(len_sq);
        else if (std::fabs(ns.x) > std::fabs(ns.y))
            dpdus = V3(-ns.z, 0.0f, ns.x) / std::sqrt(ns.x * ns.x + ns.z * ns.z);
        else
            dpdus = V3(0.0f, ns.z, -ns.y) / std::sqrt(ns.y * ns.y + ns.z * ns.z);
    }
    g.dpdus = dpdus;
    g.dpdvs = cross(ns, dpdus);
    return g;
}

static inline bool is_medium_transition(const hk_medium_interface& mi) { return mi.inside != mi.outside; }
static inline int32_t get_medium_index(const hk_medium_interface& mi, V3 wi, V3 n) { return dot(wi, n) > 0.0f ? mi.outside : mi.inside; }

// ---- shadow transmittance (intersection.jl:302-406) ----------------------------------------------
static bool trace_shadow_transmittance(const Scene& sc, V3 origin, V3 dir, float t_max, const Wavelengths& lambda, int32_t medium_idx, Spec& T_ray, Spec& r_u,
                                       Spec& r_l, Counters& cnt) {
    T_ray = Spec(1.0f);
    r_u = Spec(1.0f);
    r_l = Spec(1.0f);
    int32_t current_medium = medium_idx;
    V3 ray_o = origin;
    float t_remaining = t_max;
    for (int it = 0; it < 10; ++it) {
        if (t_remaining < 1e-6f) break;
        cnt.rays_shadow++;
        Hit h = sc.accel.closest_hit(ray_o, dir, t_remaining, &cnt.nodes, &cnt.tris);
        if (!h.hit) {
            if (current_medium >= 0) {
                Spec sT, su, sl;
                transmittance_ratio_tracking(sc.media, current_medium, ray_o, dir, t_remaining, lambda, sT, su, sl, cnt.collisions);
                T_ray = T_ray * sT;
                r_u = r_u * su;
                r_l = r_l * sl;
            }
            return true;
        }
        cnt.hits++;
        const hk_tri_meta& meta = sc.desc.meta[h.prim];
        const hk_medium_interface& mi = sc.desc.media_interfaces[meta.medium_interface_idx];
        V3 n = geometric_normal(sc, h.prim);
        bool entering = dot(dir, n) < 0.0f;
        if (!is_medium_transition(mi)) {
            float bary[3] = {1.0f - h.u - h.v, h.u, h.v};
            V2 uv = uv_barycentric(sc, h.prim, bary);
            float alpha = surface_alpha(sc.mctx, mi.material, uv);
            if (alpha < 1.0f) {
                PCG32 rng = pcg32_init(pbrt_hash(ray_o), pbrt_hash(dir));
                float au = pcg32_uniform_f32(rng);
                if (au > alpha) {
                    if (current_medium >= 0) {
                        Spec sT, su, sl;
                        transmittance_ratio_tracking(sc.media, current_medium, ray_o, dir, h.t, lambda, sT, su, sl, cnt.collisions);
                        T_ray = T_ray * sT;
                        r_u = r_u * su;
                        r_l = r_l * sl;
                    }
                    ray_o = ray_o + dir * (h.t + 1e-4f);
                    t_remaining = t_remaining - h.t - 1e-4f;
                    continue;
                }
            }
            T_ray = Spec(0.0f);
            r_u = Spec(1.0f);
            r_l = Spec(1.0f);
            return false;
        }
        if (current_medium >= 0) {
            Spec sT, su, sl;
            transmittance_ratio_tracking(sc.media, current_medium, ray_o, dir, h.t, lambda, sT, su, sl, cnt.collisions);
            T_ray = T_ray * sT;
            r_u = r_u * su;
            r_l = r_l * sl;
        }
        if (is_black(T_ray)) return true;
        current_medium = entering ? mi.inside : mi.outside;
        ray_o = ray_o + dir * (h.t + 1e-4f);
        t_remaining = t_remaining - h.t - 1e-4f;
    }
    T_ray = Spec(0.0f);
    r_u = Spec(1.0f);
    r_l = Spec(1.0f);
    return false;
}

// ---- camera medium detection (intersection.jl:690-747) -----------------------------------------
static int32_t detect_camera_medium(const Scene& sc, V3 camera_pos, Counters& cnt) {
    V3 d(0.57735027f, 0.57735027f, 0.57735027f);
    V3 o = camera_pos;
    for (int it = 0; it < 16; ++it) {
        cnt.rays_closest++;
        Hit h = sc.accel.closest_hit(o, d, INF_F, &cnt.nodes, &cnt.tris);
        if (!h.hit) return -1;
        const hk_medium_interface& mi = sc.desc.media_interfaces[sc.desc.meta[h.prim].medium_interface_idx];
        V3 n = geometric_normal(sc, h.prim);
        if (is_medium_transition(mi)) return get_medium_index(mi, -d, n);
        V3 pi = o + d * h.t;
        V3 off = dot(d, n) > 0.0f ? n : -n;
        o = pi + off * 1e-4f;
    }
    return -1;
}


struct MediumSampleItem {  // VPMediumSampleWorkItem (workitems.jl) — filled by K3 for rays inside a medium
    V3 o, d;
    float time, t_max;
    int32_t depth;
    Wavelengths lambda;
    int32_t pixel_index;
    Spec beta, r_u, r_l;
    float eta_scale;
    bool specular_bounce, any_non_specular;
    int32_t medium;
    bool has_surface_hit;
};
struct ScatterItem {  // VPMediumScatterWorkItem (workitems.jl)
    V3 p, wo;
    float time;
    Wavelengths lambda;
    int32_t pixel_index;
    Spec beta, r_u;
    int32_t depth;
    int32_t medium;
    float g;
};

// evaluate_escaped_ray_spectral  physical-wavefront/lights.jl:408-443 (sum over every light, flat order)
static Spec evaluate_escaped(const Scene& sc, V3 ray_d, const Wavelengths& lambda) {
    Spec sum(0.0f);
    for (int32_t i = 0; i < sc.desc.n_lights; ++i) {
        const hk_light& l = sc.desc.lights[i];
        if (l.kind == HK_LIGHT_AMBIENT)
            sum = sum + l.scale * light_spectrum(sc.table, l, lambda);
        else if (l.kind == HK_LIGHT_ENVIRONMENT) {  // lights.jl:408-419: bilinear env(dir) * scale, illuminant uplift
            RGBA Le_rgb = rgba_mul(env_eval(sc.desc.envmaps[l.envmap], ray_d), RGBA(l.i_rgb[0], l.i_rgb[1], l.i_rgb[2], l.i_rgb[3]));
            sum = sum + uplift_rgb_illuminant(sc.table, Le_rgb, lambda);
        } else
            sum = sum + Spec(0.0f);
    }
    return sum;
}
// compute_env_light_pdf  physical-wavefront/lights.jl:445-467 (only EnvironmentLight contributes)
static float env_light_pdf(const Scene& sc, V3 ray_d) {
    float sum = 0.0f;
    for (int32_t i = 0; i < sc.desc.n_lights; ++i) {
        const hk_light& l = sc.desc.lights[i];
        sum = sum + (l.kind == HK_LIGHT_ENVIRONMENT ? env_pdf_li(sc.desc.envmaps[l.envmap], ray_d) : 0.0f);
    }
    return sum;
}
static LightSample sample_light_full(const Scene& sc, int32_t light_idx_1based, V3 p, const Wavelengths& lambda, V2 u) {
    return sample_light_spectral(sc.table, sc.textures, sc.desc.lights[light_idx_1based - 1], p, lambda, u);
}

struct RenderState {
    int32_t width, height;
    std::vector<float> pixel_L;       // 4N
    std::vector<float> lambda, pdf;   // 4N
    std::vector<float> filter_w;      // N
    std::vector<float> s_direct_uc, s_indirect_uc, s_rr;  // N
    std::vector<V2> s_direct_u, s_indirect_u;             // N
};


// K4 (delta-tracking.jl:79-453), K5 (medium-scatter.jl:15-138), K6 (medium-scatter.jl:148-216)
static void process_media_stage(const Scene& sc, RenderState& st, RawVec<RayItem>& rays, std::vector<uint8_t>& kind, std::vector<MediumSampleItem>& msamples,
                                RawBuf<HitItem>& hits, RawBuf<EscapedItem>& escaped, RawVec<RayItem>& next_rays, RawVec<ShadowItem>& shadow_out,
                                const hk_integrator_params& ip, std::vector<Counters>& cnts, int32_t depth) {
    (void)rays;
    (void)depth;
    const int32_t n = (int32_t)kind.size();
    std::vector<ScatterItem> scat(n);
    std::vector<uint8_t> scat_valid(n, 0);
    const int32_t max_depth = ip.max_depth;
#pragma omp parallel for schedule(dynamic, 64)
    for (int32_t i = 0; i < n; ++i) {
        if (kind[i] != 3) continue;
        const MediumSampleItem& wk = msamples[i];
#if defined(_OPENMP)
        Counters& cnt = cnts[omp_get_thread_num()];
#else
        Counters& cnt = cnts[0];
#endif
        Spec beta = wk.beta, r_u = wk.r_u, r_l = wk.r_l;
        uint64_t rng = lcg_init(wk.o, wk.d, wk.t_max);
        MajorantIter it = create_majorant_iterator(sc.media, wk.medium, wk.o, wk.d, wk.t_max, wk.lambda);
        V3 ray_d = wk.d;
        bool done = false;
        for (int segi = 0; segi < 256 && !done; ++segi) {  // sample_T_maj_loop!
            MajorantSegment seg;
            if (!majorant_next(it, seg)) break;
            // sample_segment!
            Spec sm = seg.sigma_maj;
            float sm0 = sm[0];
            if (sm0 < 1e-10f) continue;
            float t = seg.t_min;
            V3 ray_o = wk.o + ray_d * t;
            for (int k = 0; k < 1024; ++k) {
                float u = lcg_next(rng);
                float dt = -std::log(maxf(1e-10f, 1.0f - u)) / sm0;
                float t_sample = t + dt;
                if (t_sample >= seg.t_max) {
                    float dt_remain = seg.t_max - t;
                    Spec T_maj = exp(-dt_remain * sm);
                    float T0 = T_maj[0];
                    if (T0 > 1e-10f) {
                        beta = beta * T_maj / T0;
                        r_u = r_u * T_maj / T0;
                        r_l = r_l * T_maj / T0;
                    }
                    break;
                }
                Spec T_maj = exp(-dt * sm);
                V3 p = ray_o + ray_d * dt;
                cnt.collisions++;
                MediumProperties mp = sample_point(sc.media, wk.medium, p, wk.lambda);
                if (!is_black(mp.Le) && wk.depth < max_depth) {
                    float pr = sm0 * T_maj[0];
                    if (pr > 1e-10f) {
                        Spec r_e = r_u * sm * T_maj / pr;
                        if (!is_black(r_e)) {
                            Spec Le_c = beta * mp.sigma_a * T_maj * mp.Le / (pr * average(r_e));
                            float* L = &st.pixel_L[4 * (size_t)(wk.pixel_index - 1)];
                            for (int c = 0; c < 4; ++c) L[c] += Le_c[c];
                        }
                    }
                }
                float p_absorb = mp.sigma_a[0] / sm0;
                float p_scatter = mp.sigma_s[0] / sm0;
                float u_event = lcg_next(rng);
                if (u_event < p_absorb) {
                    beta = Spec(0.0f);
                    done = true;
                    break;
                } else if (u_event < p_absorb + p_scatter) {
                    if (wk.depth >= max_depth) {
                        done = true;
                        break;
                    }
                    float pdf = T_maj[0] * mp.sigma_s[0];
                    if (pdf > 1e-10f) {
                        beta = beta * T_maj * mp.sigma_s / pdf;
                        r_u = r_u * T_maj * mp.sigma_s / pdf;
                    }
                    ScatterItem& si = scat[i];
                    si.p = p;
                    si.wo = -ray_d;
                    si.time = wk.time;
                    si.lambda = wk.lambda;
                    si.pixel_index = wk.pixel_index;
                    si.beta = beta;
                    si.r_u = r_u;
                    si.depth = wk.depth;
                    si.medium = wk.medium;
                    si.g = mp.g;
                    scat_valid[i] = 1;
                    done = true;
                    break;
                } else {
                    Spec sn = sm - mp.sigma_a - mp.sigma_s;
                    sn = Spec(maxf(sn[0], 0.0f), maxf(sn[1], 0.0f), maxf(sn[2], 0.0f), maxf(sn[3], 0.0f));
                    float pdf = T_maj[0] * sn[0];
                    if (pdf > 1e-10f) {
                        beta = beta * T_maj * sn / pdf;
                        r_u = r_u * T_maj * sn / pdf;
                        r_l = r_l * T_maj * sm / pdf;
                    } else {
                        beta = Spec(0.0f);
                        done = true;
                        break;
                    }
                    t = t_sample;
                    ray_o = p;  // apply_deflection is the identity (media.jl:2039)
                    if (is_black(beta) || is_black(r_u)) {
                        done = true;
                        break;
                    }
                }
            }
        }
        if (done) {
            kind[i] = 0;
            continue;
        }
        if (is_black(beta) || is_black(r_u) || wk.depth >= max_depth) {
            kind[i] = 0;
            continue;
        }
        if (!wk.has_surface_hit) {
            EscapedItem& e = escaped[i];
            e.d = ray_d;
            e.lambda = wk.lambda;
            e.pixel_index = wk.pixel_index;
            e.beta = beta;
            e.r_u = r_u;
            e.r_l = r_l;
            e.depth = wk.depth;
            e.specular_bounce = wk.specular_bounce;
            kind[i] = 2;
        } else {
            hits[i].beta = beta;
            hits[i].r_u = r_u;
            hits[i].r_l = r_l;
            kind[i] = 1;
        }
    }
    // ---- K5: direct lighting at medium scattering events (n = 0 for the light BVH) ----
    RawVec<ShadowItem> sh(n);
    std::vector<uint8_t> sh_valid(n, 0);
    RawVec<RayItem> nr(n);
    std::vector<uint8_t> nr_valid(n, 0);
#pragma omp parallel for schedule(static)
    for (int32_t i = 0; i < n; ++i) {
        if (!scat_valid[i]) continue;
        const ScatterItem& wk = scat[i];
        int32_t p0 = wk.pixel_index - 1;
        if (sc.desc.n_lights >= 1) {
            V2 u_light = st.s_direct_u[p0];
            float light_select = st.s_direct_uc[p0];
            float light_pmf;
            int32_t light_idx = sc.sampler.sample(wk.p, V3(0.0f), light_select, light_pmf);
            if (!(light_idx < 1 || light_idx > sc.desc.n_lights || light_pmf <= 0.0f)) {
                LightSample ls = sample_light_full(sc, light_idx, wk.p, wk.lambda, u_light);
                if (ls.pdf > 0.0f && !is_black(ls.Li)) {
                    float cos_t = dot(wk.wo, ls.wi);
                    float phase_val = hg_p(wk.g, cos_t);
                    if (phase_val > 0.0f) {
                        ShadowItem& s = sh[i];
                        s.Ld = wk.beta * phase_val * ls.Li;
                        float light_pdf = ls.pdf * light_pmf;
                        float phase_pdf = ls.is_delta ? 0.0f : phase_val;
                        s.r_u = wk.r_u * phase_pdf;
                        s.r_l = wk.r_u * light_pdf;
                        s.o = wk.p;
                        s.d = ls.wi;
                        s.t_max = ls.is_delta ? norm(ls.p_light - wk.p) - 0.001f : 1.0e6f;
                        s.lambda = wk.lambda;
                        s.pixel_index = wk.pixel_index;
                        s.medium = wk.medium;
                        sh_valid[i] = 1;
                    }
                }
            }
        }
        // ---- K6: phase-function sampling ----
        int32_t new_depth = wk.depth + 1;
        if (new_depth >= max_depth) continue;
        V2 u = st.s_indirect_u[p0];
        float phase_pdf;
        V3 wi = sample_hg(wk.g, wk.wo, u, phase_pdf);
        if (phase_pdf > 0.0f) {
            RayItem& r = nr[i];
            r.o = wk.p;
            r.d = wi;
            r.t_max = INF_F;
            r.time = wk.time;
            r.depth = new_depth;
            r.lambda = wk.lambda;
            r.pixel_index = wk.pixel_index;
            r.beta = wk.beta;
            r.r_u = wk.r_u;
            r.r_l = wk.r_u / phase_pdf;
            r.eta_scale = 1.0f;
            r.specular_bounce = false;
            r.any_non_specular = true;
            r.medium = wk.medium;
            nr_valid[i] = 1;
        }
    }
    append_compact(shadow_out, sh.p, sh_valid.data(), n);
    append_compact(next_rays, nr.p, nr_valid.data(), n);
}

static void render_one_sample(const Scene& sc, const hk_integrator_params& ip, const FilterParams& fp, const FilterSampler& fs, const hk_camera& cam,
                              const SobolRNG& rng, RenderState& st, int32_t sample_idx, double* pixel_rgb, double* pixel_w, bool f64, float* rgb32, float* w32,
                              Counters& cnt_total) {
    const int32_t W = st.width, H = st.height, N = W * H;
    int nthreads = 1;
#if defined(_OPENMP)
    nthreads = omp_get_max_threads();
#endif
    std::vector<Counters> cnts(nthreads);
    auto tid = []() {
#if defined(_OPENMP)
        return omp_get_thread_num();
#else
        return 0;
#endif
    };
    // initial medium
    V3 cam_pos = xform_point(cam.camera_to_world, V3(0.0f));
    int32_t initial_medium = detect_camera_medium(sc, cam_pos, cnts[0]);
    std::fill(st.pixel_L.begin(), st.pixel_L.end(), 0.0f);

    // ---- K1 ----
    RawVec<RayItem> rays(N);
    std::vector<uint8_t> valid(N, 0);
#pragma omp parallel for schedule(static)
    for (int32_t idx = 1; idx <= N; ++idx) {
        int32_t pixel_idx = idx - 1;
        int32_t x = pixel_idx % W + 1, y = pixel_idx / W + 1;
        float wavelength_u = sample_1d(rng, x, y, sample_idx, 1);
        V2 jit = sample_2d(rng, x, y, sample_idx, 3);
        float time_u = sample_1d(rng, x, y, sample_idx, 4);
        V2 lens = sample_2d(rng, x, y, sample_idx, 6);
        FilterSample f = filter_sample(fp, fs, jit);
        st.filter_w[pixel_idx] = f.weight;
        Wavelengths lam = sample_wavelengths_visible(wavelength_u);
        for (int k = 0; k < 4; ++k) {
            st.lambda[4 * pixel_idx + k] = lam.lambda[k];
            st.pdf[4 * pixel_idx + k] = lam.pdf[k];
        }
        V2 p_film((float)x + 0.5f + f.px, (float)H - (float)y + 1.0f + 0.5f + f.py);
        CamRay cr = generate_ray(cam, p_film, lens, time_u);
        RayItem& r = rays[pixel_idx];
        r.o = cr.o;
        r.d = cr.d;
        r.t_max = INF_F;
        r.time = cr.time;
        r.depth = 0;
        r.lambda = lam;
        r.pixel_index = idx;
        r.beta = Spec(1.0f);
        r.r_u = Spec(1.0f);
        r.r_l = Spec(1.0f);
        r.eta_scale = 1.0f;
        r.specular_bounce = false;
        r.any_non_specular = false;
        r.medium = initial_medium;
        valid[pixel_idx] = 1;
    }

    const bool have_lights = sc.desc.n_lights > 0;
    const bool have_media = sc.desc.n_media > 0;
    for (int32_t depth = 0; depth < ip.max_depth; ++depth) {
        const int32_t n_rays = (int32_t)rays.size();
        if (n_rays == 0) break;
        // ---- K2 ----
#pragma omp parallel for schedule(static)
        for (int32_t i = 0; i < n_rays; ++i) {
            int32_t pix = rays[i].pixel_index;
            int32_t p0 = pix - 1;
            int32_t px = p0 % W + 1, py = p0 / W + 1;
            int32_t base = 6 + 7 * depth;
            st.s_direct_uc[p0] = sample_1d(rng, px, py, sample_idx, base + 1);
            st.s_direct_u[p0] = sample_2d(rng, px, py, sample_idx, base + 3);
            st.s_indirect_uc[p0] = sample_1d(rng, px, py, sample_idx, base + 4);
            st.s_indirect_u[p0] = sample_2d(rng, px, py, sample_idx, base + 6);
            st.s_rr[p0] = sample_1d(rng, px, py, sample_idx, base + 7);
        }
        // ---- K3 trace ----
        RawBuf<HitItem> hits(n_rays);
        RawBuf<EscapedItem> escaped(n_rays);
        std::vector<MediumSampleItem> medium_samples(have_media ? n_rays : 0);
        std::vector<uint8_t> kind(n_rays, 0);  // 0 none, 1 hit, 2 escaped, 3 medium
#pragma omp parallel for schedule(dynamic, 256)
        for (int32_t i = 0; i < n_rays; ++i) {
            const RayItem& wk = rays[i];
            Counters& cnt = cnts[tid()];
            auto fill_hit = [&](HitItem& hi, const Hit& h, V3 ro, V3 rd) {
                const hk_tri_meta& meta = sc.desc.meta[h.prim];
                const hk_medium_interface& mi = sc.desc.media_interfaces[meta.medium_interface_idx];
                float bary[3] = {1.0f - h.u - h.v, h.u, h.v};
                hi.ray_o = wk.o;
                hi.ray_d = wk.d;
                hi.ray_time = wk.time;
                hi.g = surface_geometry(sc, h.prim, bary, ro, rd, h.t);
                hi.material = mi.material;
                hi.iface = mi;
                hi.face_idx = meta.primitive_index;
                hi.bary[0] = bary[0];
                hi.bary[1] = bary[1];
                hi.bary[2] = bary[2];
                hi.arealight = meta.arealight_flat_idx_1based;
                hi.triangle_area = triangle_area(sc, h.prim);
                hi.lambda = wk.lambda;
                hi.pixel_index = wk.pixel_index;
                hi.beta = wk.beta;
                hi.r_u = wk.r_u;
                hi.r_l = wk.r_l;
                hi.depth = wk.depth;
                hi.eta_scale = wk.eta_scale;
                hi.specular_bounce = wk.specular_bounce;
                hi.any_non_specular = wk.any_non_specular;
                hi.current_medium = wk.medium;
                hi.t_hit = h.t;
            };
            if (wk.medium >= 0) {
                cnt.rays_closest++;
                Hit h = sc.accel.closest_hit(wk.o, wk.d, wk.t_max, &cnt.nodes, &cnt.tris);
                MediumSampleItem& ms = medium_samples[i];
                ms.o = wk.o;
                ms.d = wk.d;
                ms.time = wk.time;
                ms.depth = wk.depth;
                ms.lambda = wk.lambda;
                ms.pixel_index = wk.pixel_index;
                ms.beta = wk.beta;
                ms.r_u = wk.r_u;
                ms.r_l = wk.r_l;
                ms.eta_scale = wk.eta_scale;
                ms.specular_bounce = wk.specular_bounce;
                ms.any_non_specular = wk.any_non_specular;
                ms.medium = wk.medium;
                ms.has_surface_hit = h.hit;
                ms.t_max = h.hit ? h.t : INF_F;
                if (h.hit) {
                    cnt.hits++;
                    fill_hit(hits[i], h, wk.o, wk.d);
                }
                kind[i] = 3;
                continue;
            }
            V3 ro = wk.o, rd = wk.d;
            for (int it = 0; it < 16; ++it) {
                cnt.rays_closest++;
                // alpha-skipped segments restart with a default Ray (t_max = Inf)
                Hit h = sc.accel.closest_hit(ro, rd, it == 0 ? wk.t_max : INF_F, &cnt.nodes, &cnt.tris);
                if (!h.hit) {
                    EscapedItem& e = escaped[i];
                    e.d = wk.d;
                    e.lambda = wk.lambda;
                    e.pixel_index = wk.pixel_index;
                    e.beta = wk.beta;
                    e.r_u = wk.r_u;
                    e.r_l = wk.r_l;
                    e.depth = wk.depth;
                    e.specular_bounce = wk.specular_bounce;
                    kind[i] = 2;
                    break;
                }
                cnt.hits++;
                const hk_tri_meta& meta = sc.desc.meta[h.prim];
                const hk_medium_interface& mi = sc.desc.media_interfaces[meta.medium_interface_idx];
                float bary[3] = {1.0f - h.u - h.v, h.u, h.v};
                V2 uv = uv_barycentric(sc, h.prim, bary);
                float alpha = surface_alpha(sc.mctx, mi.material, uv);
                if (alpha < 1.0f) {
                    PCG32 prng = pcg32_init(pbrt_hash(ro), pbrt_hash(rd));
                    float au = pcg32_uniform_f32(prng);
                    if (au > alpha) {
                        V3 pi = ro + rd * h.t;
                        V3 n = geometric_normal(sc, h.prim);
                        V3 off = dot(rd, n) > 0.0f ? n : -n;
                        ro = pi + off * 1e-4f;
                        continue;
                    }
                }
                fill_hit(hits[i], h, ro, rd);
                kind[i] = 1;
                break;
            }
        }
        // ---- K4-K6 media (delta tracking, scatter NEE, phase sampling) ----
        RawVec<RayItem> next_rays_media;
        RawVec<ShadowItem> shadow_media;
        if (have_media) {
            process_media_stage(sc, st, rays, kind, medium_samples, hits, escaped, next_rays_media, shadow_media, ip, cnts, depth);
        }
        // ---- K7 escaped ----
        if (have_lights) {
#pragma omp parallel for schedule(static)
            for (int32_t i = 0; i < n_rays; ++i) {
                if (kind[i] != 2) continue;
                const EscapedItem& wk = escaped[i];
                Spec Le = evaluate_escaped(sc, wk.d, wk.lambda);
                Spec contribution = wk.beta * Le;
                if (is_black(contribution)) continue;
                Spec final_c;
                if (wk.depth == 0 || wk.specular_bounce)
                    final_c = contribution / average(wk.r_u);
                else {
                    int32_t nl = sc.desc.n_lights;
                    float choice = nl > 0 ? 1.0f / (float)nl : 0.0f;
                    float light_pdf = env_light_pdf(sc, wk.d);
                    Spec rl = wk.r_l * choice * light_pdf;
                    Spec rsum = wk.r_u + rl;
                    float den = average(rsum);
                    final_c = den > 1e-10f ? contribution / den : contribution / average(wk.r_u);
                }
                float* L = &st.pixel_L[4 * (size_t)(wk.pixel_index - 1)];
                for (int k = 0; k < 4; ++k) L[k] += final_c[k];
            }
        }
        // ---- K8 surface hits ----
        RawBuf<MatItem> mats(n_rays);
        std::vector<uint8_t> mat_valid(n_rays, 0);
#pragma omp parallel for schedule(static)
        for (int32_t i = 0; i < n_rays; ++i) {
            if (kind[i] != 1) continue;
            const HitItem& wk = hits[i];
            Counters& cnt = cnts[tid()];
            cnt.vertices++;
            V3 wo = -wk.ray_d;
            int32_t material_idx = resolve_mix_material(sc.mctx, wk.material, wk.g.pi, wo, wk.g.uv);
            if (wk.arealight > 0) {
                const hk_light& light = sc.desc.lights[wk.arealight - 1];
                Spec Le = arealight_Le(sc.table, sc.textures, light, wo, wk.g.n, wk.g.uv, wk.lambda);
                if (!is_black(Le)) {
                    Spec contribution = wk.beta * Le;
                    Spec final_c;
                    if (wk.depth == 0 || wk.specular_bounce)
                        final_c = contribution / average(wk.r_u);
                    else {
                        float choice = sc.sampler.pmf(wk.g.pi, wk.g.n, (int32_t)wk.arealight);
                        float cos_theta = std::fabs(dot(wk.g.n, normalize(wk.ray_d)));
                        float lightPDF = 0.0f;
                        if (cos_theta > 0.0f && wk.triangle_area > 0.0f) {
                            float pdf_li = (wk.t_hit * wk.t_hit) / (cos_theta * wk.triangle_area);
                            lightPDF = choice * pdf_li;
                        }
                        Spec rl = wk.r_l * lightPDF;
                        float den = average(wk.r_u + rl);
                        final_c = den > 1e-10f ? contribution / den : contribution / average(wk.r_u);
                    }
                    float* L = &st.pixel_L[4 * (size_t)(wk.pixel_index - 1)];
                    for (int k = 0; k < 4; ++k) L[k] += final_c[k];
                }
            }
            MatItem& m = mats[i];
            m.g = wk.g;
            m.face_idx = wk.face_idx;
            m.bary[0] = wk.bary[0], m.bary[1] = wk.bary[1], m.bary[2] = wk.bary[2];
            m.wo = wo;
            m.material = material_idx;
            m.iface = wk.iface;
            m.lambda = wk.lambda;
            m.pixel_index = wk.pixel_index;
            m.beta = wk.beta;
            m.r_u = wk.r_u;
            m.r_l = wk.r_l;
            m.depth = wk.depth;
            m.eta_scale = wk.eta_scale;
            m.specular_bounce = wk.specular_bounce;
            m.any_non_specular = wk.any_non_specular;
            m.current_medium = wk.current_medium;
            mat_valid[i] = 1;
        }
        // ---- K9 direct lighting ----
        RawBuf<ShadowItem> shadows(n_rays);
        std::vector<uint8_t> shadow_valid(n_rays, 0);
        if (have_lights) {
#pragma omp parallel for schedule(static)
            for (int32_t i = 0; i < n_rays; ++i) {
                if (!mat_valid[i]) continue;
                const MatItem& wk = mats[i];
                int32_t p0 = wk.pixel_index - 1;
                V2 u_light = st.s_direct_u[p0];
                float light_select = st.s_direct_uc[p0];
                float light_pmf;
                int32_t light_idx = sc.sampler.sample(wk.g.pi, wk.g.ns, light_select, light_pmf);
                if (light_idx < 1 || light_idx > sc.desc.n_lights || light_pmf <= 0.0f) continue;
                LightSample ls = sample_light_full(sc, light_idx, wk.g.pi, wk.lambda, u_light);
                if (!(ls.pdf > 0.0f && !is_black(ls.Li))) continue;
                float bsdf_pdf;
                Spec bsdf_f = eval_bsdf_all(sc.mctx, wk.material, wk.wo, ls.wi, wk.g.ns, TexCtx(wk.g.uv, wk.face_idx, wk.bary), wk.lambda, bsdf_pdf);
                if (is_black(bsdf_f)) continue;
                // compute_direct_lighting_spectral  lights.jl:535-600
                float cos_theta = std::fabs(dot(ls.wi, wk.g.ns));
                Spec Ld = wk.beta * bsdf_f * ls.Li * cos_theta;
                if (is_black(Ld)) continue;
                V3 offset = 1e-4f * wk.g.ns;
                V3 ray_origin = dot(ls.wi, wk.g.ns) > 0.0f ? wk.g.pi + offset : wk.g.pi - offset;
                V3 to_light = ls.p_light - ray_origin;
                float t_max = std::sqrt(dot(to_light, to_light)) - 1e-3f;
                float new_bsdf_pdf = ls.is_delta ? 0.0f : bsdf_pdf;
                ShadowItem& s = shadows[i];
                s.o = ray_origin;
                s.d = ls.wi;
                s.t_max = t_max;
                s.lambda = wk.lambda;
                s.Ld = Ld;
                s.r_u = wk.r_u * new_bsdf_pdf;
                s.r_l = (wk.r_u * ls.pdf) * light_pmf;
                s.pixel_index = wk.pixel_index;
                s.medium = wk.current_medium;
                shadow_valid[i] = 1;
            }
        }
        // ---- K10 shadow rays (medium-scatter shadow items first: the reference pushes them in K5) ----
        {
            const int32_t nm = (int32_t)shadow_media.size();
#pragma omp parallel for schedule(dynamic, 256)
            for (int32_t i = 0; i < nm + n_rays; ++i) {
                const ShadowItem* wkp;
                if (i < nm)
                    wkp = &shadow_media[i];
                else {
                    if (!shadow_valid[i - nm]) continue;
                    wkp = &shadows[i - nm];
                }
                const ShadowItem& wk = *wkp;
                Counters& cnt = cnts[tid()];
                Spec T, tu, tl;
                bool visible = trace_shadow_transmittance(sc, wk.o, wk.d, wk.t_max, wk.lambda, wk.medium, T, tu, tl, cnt);
                if (visible && !is_black(T)) {
                    Spec mis = wk.r_u * tu + wk.r_l * tl;
                    float den = average(mis);
                    if (den > 1e-10f) {
                        Spec final_L = wk.Ld * T / den;
                        if (!is_black(final_L)) {
                            float* L = &st.pixel_L[4 * (size_t)(wk.pixel_index - 1)];
                            for (int k = 0; k < 4; ++k) L[k] += final_L[k];
                        }
                    }
                }
            }
        }
        // ---- K11 evaluate materials ----
        RawBuf<RayItem> out(n_rays);
        std::vector<uint8_t> out_valid(n_rays, 0);
#pragma omp parallel for schedule(static)
        for (int32_t i = 0; i < n_rays; ++i) {
            if (!mat_valid[i]) continue;
            const MatItem& wk = mats[i];
            int32_t new_depth = wk.depth + 1;
            if (new_depth >= ip.max_depth) continue;
            int32_t p0 = wk.pixel_index - 1;
            V2 u = st.s_indirect_u[p0];
            float uc = st.s_indirect_uc[p0];
            float rr = st.s_rr[p0];
            bool regularize = ip.regularize && wk.any_non_specular;
            BSDFSample s = sample_bsdf_all(sc.mctx, wk.material, wk.wo, wk.g.ns, TexCtx(wk.g.uv, wk.face_idx, wk.bary), wk.lambda, u, uc, regularize);
            if (!(s.pdf > 0.0f && !is_black(s.f))) continue;
            float cos_theta = std::fabs(dot(s.wi, wk.g.ns));
            Spec new_beta = s.is_specular ? wk.beta * s.f : wk.beta * s.f * cos_theta / s.pdf;
            float new_eta_scale = wk.eta_scale * s.eta_scale;
            Spec new_r_l = s.is_specular ? wk.r_u : wk.r_u / s.pdf;
            // russian_roulette_spectral  material-dispatch.jl:263-287 (min_depth = 3)
            Spec final_beta = new_beta;
            if (new_depth > 3) {
                float q = maxf(0.05f, 1.0f - max_component(new_beta));
                if (rr < q) continue;
                final_beta = new_beta * (1.0f / (1.0f - q));
            }
            int32_t new_medium = is_medium_transition(wk.iface) ? get_medium_index(wk.iface, s.wi, wk.g.n) : wk.current_medium;
            V3 off = dot(s.wi, wk.g.n) > 0.0f ? wk.g.n : -wk.g.n;
            RayItem& r = out[i];
            r.o = wk.g.pi + off * 0.0001f;
            r.d = s.wi;
            r.t_max = INF_F;
            r.time = 0.0f;
            r.depth = new_depth;
            r.lambda = wk.lambda;
            r.pixel_index = wk.pixel_index;
            r.beta = final_beta;
            r.r_u = wk.r_u;
            r.r_l = new_r_l;
            r.eta_scale = new_eta_scale;
            r.specular_bounce = s.is_specular;
            r.any_non_specular = wk.any_non_specular || !s.is_specular;
            r.medium = new_medium;
            out_valid[i] = 1;
        }
        RawVec<RayItem> next;
        next.reserve(n_rays + next_rays_media.size());
        for (auto& r : next_rays_media) next.push_back(r);
        append_compact(next, out.p, out_valid.data(), n_rays);
        rays.swap(next);
    }

    // ---- K12 ----
#pragma omp parallel for schedule(static)
    for (int32_t p = 0; p < N; ++p) {
        Spec L(st.pixel_L[4 * p], st.pixel_L[4 * p + 1], st.pixel_L[4 * p + 2], st.pixel_L[4 * p + 3]);
        Wavelengths lam;
        for (int k = 0; k < 4; ++k) {
            lam.lambda[k] = st.lambda[4 * p + k];
            lam.pdf[k] = st.pdf[4 * p + k];
        }
        V3 xyz = spectral_to_xyz(sc.cie, L, lam);
        V3 rgb = xyz_to_linear_srgb(xyz);
        rgb = V3(maxf(0.0f, rgb.x), maxf(0.0f, rgb.y), maxf(0.0f, rgb.z));
        float m = maxf(maxf(rgb.x, rgb.y), rgb.z);
        if (m > ip.max_component_value) rgb = rgb * (ip.max_component_value / m);
        float wgt = st.filter_w[p];
        if (f64) {
            pixel_rgb[3 * p] += (double)(wgt * rgb.x);
            pixel_rgb[3 * p + 1] += (double)(wgt * rgb.y);
            pixel_rgb[3 * p + 2] += (double)(wgt * rgb.z);
            pixel_w[p] += (double)wgt;
        } else {
            rgb32[3 * p] += wgt * rgb.x;
            rgb32[3 * p + 1] += wgt * rgb.y;
            rgb32[3 * p + 2] += wgt * rgb.z;
            w32[p] += wgt;
        }
    }
    for (auto& c : cnts) {
        cnt_total.rays_closest += c.rays_closest;
        cnt_total.rays_shadow += c.rays_shadow;
        cnt_total.nodes += c.nodes;
        cnt_total.tris += c.tris;
        cnt_total.hits += c.hits;
        cnt_total.vertices += c.vertices;
        cnt_total.collisions += c.collisions;
    }
}

}  // namespace hko

// =====================================================================================================
// C entry points (loaded by tests / smoke / bench cpu_baseline through ctypes)
// =====================================================================================================
using namespace hko;

struct hko_scene {
    Scene sc;
};

extern "C" {

int32_t hko_scene_create(const hk_scene_desc* desc, const hk_tables* tables, hko_scene** out) {
    hko_scene* s = new hko_scene();
    Scene& sc = s->sc;
    sc.desc = *desc;  // borrowed pointers: the caller keeps the arrays alive for the scene's lifetime
    sc.accel.build(desc->positions, desc->n_triangles);
    sc.sampler.build(desc->lights, desc->n_lights);
    sc.table.res = tables->rgb2spec_res;
    sc.table.scale = tables->rgb2spec_scale;
    sc.table.coeffs = tables->rgb2spec_coeffs;
    sc.cie.x = tables->cie_x;
    sc.cie.y = tables->cie_y;
    sc.cie.z = tables->cie_z;
    sc.sobol = tables->sobol_matrices;
    sc.textures.tex = desc->textures;
    sc.textures.n = desc->n_textures;
    sc.textures.envmaps = desc->envmaps;
    sc.mctx.table = &sc.table;
    sc.mctx.textures = sc.textures;
    sc.mctx.materials = desc->materials;
    sc.mctx.n_materials = desc->n_materials;
    sc.mctx.spectra = desc->spectra;
    init_media(sc.media, desc, &sc.table);
    *out = s;
    return 0;
}
int32_t hko_scene_destroy(hko_scene* s) {
    delete s;
    return 0;
}

// Render samples first..first+n-1 (stride `stride`) on top of accum = [rgb 3N | weight N] (f32 or f64).
int32_t hko_render(hko_scene* s, const hk_integrator_params* ip, const hk_camera* cam, int32_t width, int32_t height, int32_t first_sample_idx,
                   int32_t n_samples, int32_t stride, void* accum, hk_stats* stats) {
    Scene& sc = s->sc;
    FilterParams fp = make_filter_params(*ip);
    FilterSampler fs = build_filter_sampler(fp);
    int spp = ip->samples_per_pixel > 4096 ? ip->samples_per_pixel : 4096;  // volpath.jl:475
    SobolRNG rng = make_sobol_rng(sc.sobol, ip->sampler_seed, width, height, spp);
    RenderState st;
    st.width = width;
    st.height = height;
    size_t N = (size_t)width * height;
    st.pixel_L.assign(4 * N, 0.0f);
    st.lambda.assign(4 * N, 0.0f);
    st.pdf.assign(4 * N, 0.0f);
    st.filter_w.assign(N, 0.0f);
    st.s_direct_uc.assign(N, 0.0f);
    st.s_indirect_uc.assign(N, 0.0f);
    st.s_rr.assign(N, 0.0f);
    st.s_direct_u.assign(N, V2());
    st.s_indirect_u.assign(N, V2());
    bool f64 = ip->accumulate_f64 != 0;
    Counters cnt;
    for (int32_t k = 0; k < n_samples; ++k) {
        int32_t sidx = first_sample_idx + k * stride;
        if (f64)
            render_one_sample(sc, *ip, fp, fs, *cam, rng, st, sidx, (double*)accum, (double*)accum + 3 * N, true, nullptr, nullptr, cnt);
        else
            render_one_sample(sc, *ip, fp, fs, *cam, rng, st, sidx, nullptr, nullptr, false, (float*)accum, (float*)accum + 3 * N, cnt);
    }
    if (stats) {
        stats->rays_closest += cnt.rays_closest;
        stats->rays_shadow += cnt.rays_shadow;
        stats->bvh_nodes_visited += cnt.nodes;
        stats->tris_tested += cnt.tris;
        stats->hits_accepted += cnt.hits;
        stats->path_vertices += cnt.vertices;
        stats->medium_collisions += cnt.collisions;
        stats->light_bvh_nodes = sc.sampler.nodes_evaluated;
    }
    return 0;
}

// K13: framebuffer[py,px] = rgb/weight as Julia Matrix{RGB{Float32}}[height,width] (column-major)
int32_t hko_finalize(int32_t width, int32_t height, int32_t f64, const void* accum, float* out_hw3) {
    size_t N = (size_t)width * height;
    for (size_t p = 0; p < N; ++p) {
        int32_t px = (int32_t)(p % width), py = (int32_t)(p / width);
        float r, g, b;
        if (f64) {
            const double* a = (const double*)accum;
            double w = a[3 * N + p];
            if (w > 0.0) {
                double inv = 1.0 / w;
                r = (float)(a[3 * p] * inv);
                g = (float)(a[3 * p + 1] * inv);
                b = (float)(a[3 * p + 2] * inv);
            } else
                r = g = b = 0.0f;
        } else {
            const float* a = (const float*)accum;
            float w = a[3 * N + p];
            if (w > 0.0f) {
                float inv = 1.0f / w;
                r = a[3 * p] * inv;
                g = a[3 * p + 1] * inv;
                b = a[3 * p + 2] * inv;
            } else
                r = g = b = 0.0f;
        }
        float* o = out_hw3 + 3 * ((size_t)py + (size_t)height * px);
        o[0] = r;
        o[1] = g;
        o[2] = b;
    }
    return 0;
}

int32_t hko_trace_closest(hko_scene* s, int32_t n, const float* o3, const float* d3, const float* tmax, float* out_t, int32_t* out_prim, float* out_uv2) {
#pragma omp parallel for schedule(dynamic, 1024)
    for (int32_t i = 0; i < n; ++i) {
        Hit h = s->sc.accel.closest_hit(V3(o3[3 * i], o3[3 * i + 1], o3[3 * i + 2]), V3(d3[3 * i], d3[3 * i + 1], d3[3 * i + 2]), tmax[i]);
        out_t[i] = h.hit ? h.t : INF_F;
        out_prim[i] = h.hit ? h.prim : -1;
        out_uv2[2 * i] = h.hit ? h.u : 0.0f;
        out_uv2[2 * i + 1] = h.hit ? h.v : 0.0f;
    }
    return 0;
}

int32_t hko_sobol(const uint32_t* matrices, int32_t width, int32_t height, int32_t spp, uint32_t seed, int32_t n, const int32_t* px, const int32_t* py,
                  const int32_t* sample_idx, const int32_t* dim, float* out_1d, float* out_2d) {
    SobolRNG r = make_sobol_rng(matrices, seed, width, height, spp);
    for (int32_t i = 0; i < n; ++i) {
        out_1d[i] = sample_1d(r, px[i], py[i], sample_idx[i], dim[i]);
        V2 v = sample_2d(r, px[i], py[i], sample_idx[i], dim[i]);
        out_2d[2 * i] = v.x;
        out_2d[2 * i + 1] = v.y;
    }
    return 0;
}

int32_t hko_camera(const uint32_t* matrices, const hk_integrator_params* ip, const hk_camera* cam, int32_t width, int32_t height, int32_t n, const int32_t* px,
                   const int32_t* py, const int32_t* sample_idx, float* out15) {
    FilterParams fp = make_filter_params(*ip);
    FilterSampler fs = build_filter_sampler(fp);
    int spp = ip->samples_per_pixel > 4096 ? ip->samples_per_pixel : 4096;
    SobolRNG rng = make_sobol_rng(matrices, ip->sampler_seed, width, height, spp);
    for (int32_t i = 0; i < n; ++i) {
        int32_t x = px[i], y = py[i], s = sample_idx[i];
        float wavelength_u = sample_1d(rng, x, y, s, 1);
        V2 jit = sample_2d(rng, x, y, s, 3);
        float time_u = sample_1d(rng, x, y, s, 4);
        V2 lens = sample_2d(rng, x, y, s, 6);
        FilterSample f = filter_sample(fp, fs, jit);
        Wavelengths lam = sample_wavelengths_visible(wavelength_u);
        V2 p_film((float)x + 0.5f + f.px, (float)height - (float)y + 1.0f + 0.5f + f.py);
        CamRay cr = generate_ray(*cam, p_film, lens, time_u);
        float* o = out15 + 15 * (size_t)i;
        for (int k = 0; k < 4; ++k) {
            o[k] = lam.lambda[k];
            o[4 + k] = lam.pdf[k];
        }
        o[8] = f.weight;
        o[9] = cr.o.x;
        o[10] = cr.o.y;
        o[11] = cr.o.z;
        o[12] = cr.d.x;
        o[13] = cr.d.y;
        o[14] = cr.d.z;
    }
    return 0;
}

int32_t hko_uplift(const hk_tables* tables, int32_t mode, int32_t n, const float* rgb, const float* lambda, float* out) {
    RGB2SpecTable t{tables->rgb2spec_res, tables->rgb2spec_scale, tables->rgb2spec_coeffs};
    for (int32_t i = 0; i < n; ++i) {
        Wavelengths w;
        for (int k = 0; k < 4; ++k) {
            w.lambda[k] = lambda[4 * i + k];
            w.pdf[k] = 1.0f;
        }
        RGBA c(rgb[3 * i], rgb[3 * i + 1], rgb[3 * i + 2]);
        Spec s = mode == 0 ? uplift_rgb(t, c, w) : (mode == 1 ? uplift_rgb_unbounded(t, c, w) : uplift_rgb_illuminant(t, c, w));
        for (int k = 0; k < 4; ++k) out[4 * i + k] = s[k];
    }
    return 0;
}

int32_t hko_light_bvh(hko_scene* s, int32_t n, const float* p3, const float* n3, const float* u, int32_t* out_light, float* out_pmf, const int32_t* query_light,
                      float* out_query_pmf) {
    for (int32_t i = 0; i < n; ++i) {
        V3 p(p3[3 * i], p3[3 * i + 1], p3[3 * i + 2]), nn(n3[3 * i], n3[3 * i + 1], n3[3 * i + 2]);
        float pmf;
        out_light[i] = s->sc.sampler.sample(p, nn, u[i], pmf);
        out_pmf[i] = pmf;
        if (query_light && out_query_pmf) out_query_pmf[i] = s->sc.sampler.pmf(p, nn, query_light[i]);
    }
    return 0;
}
int32_t hko_light_bvh_copy(hko_scene* s, int32_t* n_nodes, float* nodes_out, uint32_t* bit_trails) {
    const LightSampler& ls = s->sc.sampler;
    *n_nodes = (int32_t)ls.nodes.size();
    if (nodes_out)
        for (size_t i = 0; i < ls.nodes.size(); ++i) {
            const LightBVHNode& nd = ls.nodes[i];
            float* o = nodes_out + 16 * i;
            o[0] = nd.bmin.x; o[1] = nd.bmin.y; o[2] = nd.bmin.z;
            o[3] = nd.bmax.x; o[4] = nd.bmax.y; o[5] = nd.bmax.z;
            o[6] = nd.w.x; o[7] = nd.w.y; o[8] = nd.w.z;
            o[9] = nd.phi; o[10] = nd.cos_o; o[11] = nd.cos_e;
            o[12] = nd.two_sided ? 1.0f : 0.0f;
            o[13] = (float)nd.child1_or_light;
            o[14] = nd.is_leaf ? 1.0f : 0.0f;
            o[15] = 0.0f;
        }
    if (bit_trails)
        for (size_t i = 0; i < ls.bit_trail.size(); ++i) bit_trails[i] = ls.bit_trail[i];
    return 0;
}

// Point-wise BSDF evaluation against a scene's material table (material-dispatch.jl:23-53), n vertices:
//   wo/wi/ns: 3 floats each, lambda: 4, u: 2, uc: 1; mode 0 = sample (out: wi3, f4, pdf, is_specular, eta_scale = 10),
//   mode 1 = evaluate (out: f4, pdf, 0... = 10).  uv = (0,0).
int32_t hko_bsdf(hko_scene* s, int32_t mode, int32_t mat_idx, int32_t regularize, int32_t n, const float* wo, const float* wi, const float* ns, const float* lambda,
                 const float* u, const float* uc, float* out) {
    Scene& sc = s->sc;
    for (int i = 0; i < n; ++i) {
        Wavelengths w;
        for (int k = 0; k < 4; ++k) w.lambda[k] = lambda[4 * i + k], w.pdf[k] = 1.0f;
        V3 o(wo[3 * i], wo[3 * i + 1], wo[3 * i + 2]), nn(ns[3 * i], ns[3 * i + 1], ns[3 * i + 2]);
        float* r = out + 10 * (size_t)i;
        for (int k = 0; k < 10; ++k) r[k] = 0.0f;
        if (mode == 0) {
            BSDFSample b = sample_bsdf_all(sc.mctx, mat_idx, o, nn, V2(0, 0), w, V2(u[2 * i], u[2 * i + 1]), uc[i], regularize != 0);
            r[0] = b.wi.x, r[1] = b.wi.y, r[2] = b.wi.z;
            for (int k = 0; k < 4; ++k) r[3 + k] = b.f[k];
            r[7] = b.pdf, r[8] = b.is_specular ? 1.0f : 0.0f, r[9] = b.eta_scale;
        } else {
            V3 d(wi[3 * i], wi[3 * i + 1], wi[3 * i + 2]);
            float pdf = 0.0f;
            Spec f = eval_bsdf_all(sc.mctx, mat_idx, o, d, nn, V2(0, 0), w, pdf);
            for (int k = 0; k < 4; ++k) r[k] = f[k];
            r[4] = pdf;
        }
    }
    return 0;
}

// Point-wise light sampling / escaped-ray evaluation (physical-wavefront/lights.jl:39-297, 408-467), n points:
//   mode 0: sample_light_spectral(light `light_idx_1based`, p, lambda, u = in3.xy) -> out[12] = wi3, pdf, Li4, p_light3, is_delta
//   mode 1: escaped ray with direction in3: out[12] = Le4 (all lights), env pdf, 0...
int32_t hko_light(hko_scene* s, int32_t mode, int32_t light_idx_1based, int32_t n, const float* p3, const float* in3, const float* lambda, float* out) {
    Scene& sc = s->sc;
    for (int i = 0; i < n; ++i) {
        Wavelengths w;
        for (int k = 0; k < 4; ++k) w.lambda[k] = lambda[4 * i + k], w.pdf[k] = 1.0f;
        float* r = out + 12 * (size_t)i;
        for (int k = 0; k < 12; ++k) r[k] = 0.0f;
        V3 a(in3[3 * i], in3[3 * i + 1], in3[3 * i + 2]);
        if (mode == 0) {
            LightSample ls = sample_light_full(sc, light_idx_1based, V3(p3[3 * i], p3[3 * i + 1], p3[3 * i + 2]), w, V2(a.x, a.y));
            r[0] = ls.wi.x, r[1] = ls.wi.y, r[2] = ls.wi.z, r[3] = ls.pdf;
            for (int k = 0; k < 4; ++k) r[4 + k] = ls.Li[k];
            r[8] = ls.p_light.x, r[9] = ls.p_light.y, r[10] = ls.p_light.z, r[11] = ls.is_delta ? 1.0f : 0.0f;
        } else {
            Spec Le = evaluate_escaped(sc, a, w);
            for (int k = 0; k < 4; ++k) r[k] = Le[k];
            r[4] = env_light_pdf(sc, a);
        }
    }
    return 0;
}

// resolve_mix_material (mix-material.jl:222-238) for n hit points
int32_t hko_mix_resolve(hko_scene* s, int32_t mat_idx, int32_t n, const float* p3, const float* wo3, const float* uv2, int32_t* out_mat) {
    for (int i = 0; i < n; ++i)
        out_mat[i] = resolve_mix_material(s->sc.mctx, mat_idx, V3(p3[3 * i], p3[3 * i + 1], p3[3 * i + 2]), V3(wo3[3 * i], wo3[3 * i + 1], wo3[3 * i + 2]),
                                          V2(uv2[2 * i], uv2[2 * i + 1]));
    return 0;
}

// Point-wise media: mode 0 = sample_point -> out[13] = sigma_a4, sigma_s4, Le4, g (through the NanoVDB TREE WALK, nanovdb.jl:315-388);
// mode 1 = create_majorant_iterator + ray_majorant_next -> out[49] = segment count (<= 256), (t_min, t_max, sigma_maj[1]) of the first 16
// K4 + K5 + K6 of n caller-supplied rays through the oracle's OWN stage code (process_media_stage above, nothing restated): ray i is
// pixel i + 1 of an n-pixel state whose Sobol draws (direct_uc, direct_u, indirect_u) the caller supplies.  in[i] = o3, d3, t_max (inf: the
// ray meets no surface), lambda4, beta4, r_u4, r_l4 (23 floats); out[i] (56 floats) = fate (0 ended, 1 reached its surface, 2 escaped),
// beta4, r_u4, r_l4 of a survivor, L4 added to the pixel by medium emission; continuation ray: valid, o3, d3, beta4, r_u4, r_l4; shadow
// ray: valid, o3, d3, t_max, Ld4, r_u4, r_l4.  Test entry: the per-ray pins of tests/test_control_flow_pin.py.
int32_t hko_media_stage(hko_scene* s, int32_t medium_idx, int32_t depth, int32_t max_depth, int32_t n, const float* in23, const float* direct_uc, const float* direct_u2,
                        const float* indirect_u2, float* out56) {
    Scene& sc = s->sc;
    if (medium_idx < 0 || medium_idx >= sc.desc.n_media) return -1;
    RenderState st;
    st.width = n, st.height = 1;
    st.pixel_L.assign(4 * (size_t)n, 0.0f);
    st.s_direct_uc.assign(direct_uc, direct_uc + n);
    st.s_direct_u.resize(n);
    st.s_indirect_u.resize(n);
    for (int i = 0; i < n; ++i) st.s_direct_u[i] = V2(direct_u2[2 * i], direct_u2[2 * i + 1]), st.s_indirect_u[i] = V2(indirect_u2[2 * i], indirect_u2[2 * i + 1]);
    RawVec<RayItem> rays;
    std::vector<uint8_t> kind(n, 3);
    std::vector<MediumSampleItem> ms(n);
    RawBuf<HitItem> hits(n);
    RawBuf<EscapedItem> escaped(n);
    RawVec<RayItem> next_rays;
    RawVec<ShadowItem> shadows;
    for (int i = 0; i < n; ++i) {
        const float* r = in23 + 23 * (size_t)i;
        MediumSampleItem& m = ms[i];
        m.o = V3(r[0], r[1], r[2]), m.d = V3(r[3], r[4], r[5]);
        m.time = 0.0f, m.t_max = r[6];
        m.depth = depth;
        for (int k = 0; k < 4; ++k) m.lambda.lambda[k] = r[7 + k], m.lambda.pdf[k] = 1.0f;
        m.pixel_index = i + 1;
        m.beta = Spec(r[11], r[12], r[13], r[14]), m.r_u = Spec(r[15], r[16], r[17], r[18]), m.r_l = Spec(r[19], r[20], r[21], r[22]);
        m.eta_scale = 1.0f, m.specular_bounce = false, m.any_non_specular = false;
        m.medium = medium_idx;
        m.has_surface_hit = std::isfinite(r[6]);
    }
    hk_integrator_params ip{};
    ip.max_depth = max_depth;
#if defined(_OPENMP)
    std::vector<Counters> cnts((size_t)omp_get_max_threads());
#else
    std::vector<Counters> cnts(1);
#endif
    process_media_stage(sc, st, rays, kind, ms, hits, escaped, next_rays, shadows, ip, cnts, depth);
    for (size_t k = 0; k < 56 * (size_t)n; ++k) out56[k] = 0.0f;
    for (int i = 0; i < n; ++i) {
        float* o = out56 + 56 * (size_t)i;
        o[0] = (float)kind[i];
        const Spec *b = nullptr, *u = nullptr, *l = nullptr;
        if (kind[i] == 1) b = &hits[i].beta, u = &hits[i].r_u, l = &hits[i].r_l;
        if (kind[i] == 2) b = &escaped[i].beta, u = &escaped[i].r_u, l = &escaped[i].r_l;
        if (b)
            for (int k = 0; k < 4; ++k) o[1 + k] = (*b)[k], o[5 + k] = (*u)[k], o[9 + k] = (*l)[k];
        for (int k = 0; k < 4; ++k) o[13 + k] = st.pixel_L[4 * (size_t)i + k];
    }
    for (size_t j = 0; j < next_rays.size(); ++j) {
        const RayItem& r = next_rays[j];
        float* o = out56 + 56 * (size_t)(r.pixel_index - 1) + 17;
        o[0] = 1.0f;
        o[1] = r.o.x, o[2] = r.o.y, o[3] = r.o.z, o[4] = r.d.x, o[5] = r.d.y, o[6] = r.d.z;
        for (int k = 0; k < 4; ++k) o[7 + k] = r.beta[k], o[11 + k] = r.r_u[k], o[15 + k] = r.r_l[k];
    }
    for (size_t j = 0; j < shadows.size(); ++j) {
        const ShadowItem& r = shadows[j];
        float* o = out56 + 56 * (size_t)(r.pixel_index - 1) + 36;
        o[0] = 1.0f;
        o[1] = r.o.x, o[2] = r.o.y, o[3] = r.o.z, o[4] = r.d.x, o[5] = r.d.y, o[6] = r.d.z, o[7] = r.t_max;
        for (int k = 0; k < 4; ++k) o[8 + k] = r.Ld[k], o[12 + k] = r.r_u[k], o[16 + k] = r.r_l[k];
    }
    return 0;
}
int32_t hko_medium(hko_scene* s, int32_t mode, int32_t medium_idx, int32_t n, const float* a3, const float* b3, const float* tmax, const float* lambda, float* out) {
    Scene& sc = s->sc;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; ++i) {
        Wavelengths w;
        for (int k = 0; k < 4; ++k) w.lambda[k] = lambda[4 * i + k], w.pdf[k] = 1.0f;
        V3 a(a3[3 * i], a3[3 * i + 1], a3[3 * i + 2]);
        if (mode == 2) {
            // the whole shadow walk of ONE ray (trace_shadow_transmittance above: intersection.jl:302-542) starting in medium `medium_idx`
            // (-1: vacuum): -> T_ray[4], r_u[4], r_l[4], visible.  Test entry: the per-ray pin of tests/test_control_flow_pin.py.
            Spec T, ru, rl;
            Counters cnt{};
            const bool vis = trace_shadow_transmittance(sc, a, V3(b3[3 * i], b3[3 * i + 1], b3[3 * i + 2]), tmax[i], w, medium_idx, T, ru, rl, cnt);
            float* r = out + 13 * (size_t)i;
            for (int k = 0; k < 4; ++k) r[k] = T[k], r[4 + k] = ru[k], r[8 + k] = rl[k];
            r[12] = vis ? 1.0f : 0.0f;
        } else if (mode == 0) {
            MediumProperties mp = sample_point(sc.media, medium_idx, a, w);
            float* r = out + 13 * (size_t)i;
            for (int k = 0; k < 4; ++k) r[k] = mp.sigma_a[k], r[4 + k] = mp.sigma_s[k], r[8 + k] = mp.Le[k];
            r[12] = mp.g;
        } else {
            float* r = out + 49 * (size_t)i;
            for (int k = 0; k < 49; ++k) r[k] = 0.0f;
            MajorantIter it = create_majorant_iterator(sc.media, medium_idx, a, V3(b3[3 * i], b3[3 * i + 1], b3[3 * i + 2]), tmax[i], w);
            MajorantSegment seg;
            int count = 0;
            while (count < 256 && majorant_next(it, seg)) {
                if (count < 16) r[1 + 3 * count] = seg.t_min, r[2 + 3 * count] = seg.t_max, r[3 + 3 * count] = seg.sigma_maj[0];
                ++count;
            }
            r[0] = (float)count;
        }
    }
    return 0;
}

// postprocess_kernel! (src/postprocess.jl:185-250) on a Julia-layout [h,w] RGB framebuffer
static float pp_unch2(float x) {
    const float A = 0.15f, B = 0.50f, C = 0.10f, D = 0.20f, E = 0.02f, F = 0.30f;
    return ((x * (A * x + C * B) + D * E) / (x * (A * x + B) + D * F)) - E / F;
}
static float pp_filmic(float x) {
    x = maxf(0.0f, x - 0.004f);
    return (x * (6.2f * x + 0.5f)) / (x * (6.2f * x + 1.7f) + 0.06f);
}
// denoise! (src/denoise.jl:301-376): variance (:236-286) + a-trous passes (:136-229), Julia [h,w] buffers; see hk_denoise.
static inline float dn_lum(float r, float g, float b) { return 0.2126f * r + 0.7152f * g + 0.0722f * b; }
int32_t hko_denoise(const hk_denoise_params* Pp, int32_t w, int32_t h, const float* src, const float* normal, const float* depth, float* dst, float* src_after) {
    const hk_denoise_params& P = *Pp;
    const long n = (long)h * w;
    std::vector<float> a(src, src + 3 * n), b(3 * n, 0.0f), var(n, 0.0f);
    if (P.use_variance) {
#pragma omp parallel for schedule(static)
        for (long i = 0; i < n; ++i) {
            int row = (int)(i % h), col = (int)(i / h);
            float sum = 0.0f, sum_sq = 0.0f;
            int count = 0;
            for (int dy = -1; dy <= 1; ++dy)
                for (int dx = -1; dx <= 1; ++dx) {
                    int qr = row + dy, qc = col + dx;
                    if (qr >= 0 && qr < h && qc >= 0 && qc < w) {
                        const float* q = &a[3 * ((long)qc * h + qr)];
                        float lum = dn_lum(q[0], q[1], q[2]);
                        sum += lum;
                        sum_sq += lum * lum;
                        ++count;
                    }
                }
            float mean = sum / (float)count, mean_sq = sum_sq / (float)count;
            var[i] = maxf(0.0f, mean_sq - mean * mean);
        }
    }
    const float K1D[5] = {1.0f / 16.0f, 1.0f / 4.0f, 3.0f / 8.0f, 1.0f / 4.0f, 1.0f / 16.0f};
    for (int it = 1; it <= P.iterations; ++it) {
        const int step = 1 << (it - 1);
        const std::vector<float>& in = (it & 1) ? a : b;
        std::vector<float>& out = (it & 1) ? b : a;
#pragma omp parallel for schedule(static)
        for (long i = 0; i < n; ++i) {
            int row = (int)(i % h), col = (int)(i / h);
            float r_p = in[3 * i], g_p = in[3 * i + 1], b_p = in[3 * i + 2];
            float lum_p = dn_lum(r_p, g_p, b_p);
            float nx = normal[3 * i], ny = normal[3 * i + 1], nz = normal[3 * i + 2];
            float d_p = depth[i];
            float var_p = P.use_variance ? var[i] : 0.0f;
            float sr = 0.0f, sg = 0.0f, sb = 0.0f, sw = 0.0f;
            for (int dyi = 0; dyi < 5; ++dyi)
                for (int dxi = 0; dxi < 5; ++dxi) {
                    int qr = row + (dyi - 2) * step, qc = col + (dxi - 2) * step;
                    qr = std::min(std::max(qr, 0), h - 1);
                    qc = std::min(std::max(qc, 0), w - 1);
                    long q = (long)qc * h + qr;
                    float r_q = in[3 * q], g_q = in[3 * q + 1], b_q = in[3 * q + 2];
                    float lum_q = dn_lum(r_q, g_q, b_q);
                    float w_spatial = K1D[dxi] * K1D[dyi];
                    float diff = std::fabs(lum_p - lum_q);
                    float eff = var_p > 0.0f ? P.sigma_color * std::sqrt(var_p) + 1.0e-4f : P.sigma_color;   // weight_color :76-88
                    float w_color = std::exp(-diff / eff);
                    float dotv = nx * normal[3 * q] + ny * normal[3 * q + 1] + nz * normal[3 * q + 2];
                    float w_norm = std::pow(maxf(0.0f, dotv), P.sigma_normal);                                    // weight_normal :96-103
                    float w_depth = std::exp(-std::fabs(d_p - depth[q]) / (P.sigma_depth * (float)step + 1.0e-4f));  // weight_depth :111-118
                    float weight = w_spatial * w_color * w_norm * w_depth;
                    sr += r_q * weight;
                    sg += g_q * weight;
                    sb += b_q * weight;
                    sw += weight;
                }
            if (sw > 1.0e-6f) {
                float inv = 1.0f / sw;
                out[3 * i] = sr * inv, out[3 * i + 1] = sg * inv, out[3 * i + 2] = sb * inv;
            } else
                out[3 * i] = r_p, out[3 * i + 1] = g_p, out[3 * i + 2] = b_p;
        }
    }
    const std::vector<float>& last = (P.iterations & 1) ? b : a;
    std::memcpy(dst, last.data(), sizeof(float) * 3 * n);
    if (src_after) std::memcpy(src_after, a.data(), sizeof(float) * 3 * n);
    return 0;
}
int32_t hko_postprocess(const hk_postprocess_params* Pp, int32_t w, int32_t h, const float* src, const float* depth, float* dst) {
    const hk_postprocess_params& P = *Pp;
    const long n = (long)h * w;
    for (long i = 0; i < n; ++i) {
        float r = src[3 * i] * P.exposure, g = src[3 * i + 1] * P.exposure, b = src[3 * i + 2] * P.exposure;
        if (P.apply_wb) {
            float ro = P.wb[0] * r + P.wb[1] * g + P.wb[2] * b, go = P.wb[3] * r + P.wb[4] * g + P.wb[5] * b, bo = P.wb[6] * r + P.wb[7] * g + P.wb[8] * b;
            r = maxf(0.0f, ro), g = maxf(0.0f, go), b = maxf(0.0f, bo);
        }
        r = r * P.imaging_ratio, g = g * P.imaging_ratio, b = b * P.imaging_ratio;
        if (P.tonemap == HK_TONEMAP_REINHARD || P.tonemap == HK_TONEMAP_REINHARD_EXT) {
            float lum = 0.2126f * r + 0.7152f * g + 0.0722f * b;
            float sc = 1.0f;
            if (lum > 0.0f) sc = P.tonemap == HK_TONEMAP_REINHARD ? 1.0f / (1.0f + lum) : (1.0f + lum / (P.white_point * P.white_point)) / (1.0f + lum);
            r = clampf(r * sc, 0.0f, 1.0f), g = clampf(g * sc, 0.0f, 1.0f), b = clampf(b * sc, 0.0f, 1.0f);
        } else if (P.tonemap == HK_TONEMAP_ACES) {
            const float a = 2.51f, bc = 0.03f, c = 2.43f, d = 0.59f, e = 0.14f;
            r = clampf((r * (a * r + bc)) / (r * (c * r + d) + e), 0.0f, 1.0f);
            g = clampf((g * (a * g + bc)) / (g * (c * g + d) + e), 0.0f, 1.0f);
            b = clampf((b * (a * b + bc)) / (b * (c * b + d) + e), 0.0f, 1.0f);
        } else if (P.tonemap == HK_TONEMAP_UNCHARTED2) {
            float ws = 1.0f / pp_unch2(11.2f);
            r = clampf(pp_unch2(r * 2.0f) * ws, 0.0f, 1.0f), g = clampf(pp_unch2(g * 2.0f) * ws, 0.0f, 1.0f), b = clampf(pp_unch2(b * 2.0f) * ws, 0.0f, 1.0f);
        } else if (P.tonemap == HK_TONEMAP_FILMIC) {
            r = pp_filmic(r), g = pp_filmic(g), b = pp_filmic(b);
        } else
            r = clampf(r, 0.0f, 1.0f), g = clampf(g, 0.0f, 1.0f), b = clampf(b, 0.0f, 1.0f);
        if (P.apply_gamma) r = std::pow(r, P.inv_gamma), g = std::pow(g, P.inv_gamma), b = std::pow(b, P.inv_gamma);
        if (P.mask_escaped && depth) {
            int row = (int)(i % h) + 1, col = (int)(i / h) + 1, d_row = h - row + 1;
            int escaped = 0, total = 0;
            for (int dr = -1; dr <= 1; ++dr)
                for (int dc = -1; dc <= 1; ++dc) {
                    int nr = d_row + dr, nc = col + dc;
                    if (nr >= 1 && nr <= h && nc >= 1 && nc <= w) {
                        escaped += std::isinf(depth[(long)(nc - 1) * h + nr - 1]) ? 1 : 0;
                        total += 1;
                    }
                }
            float alpha = (float)escaped / (float)total;
            r = r * (1.0f - alpha) + P.bg[0] * alpha, g = g * (1.0f - alpha) + P.bg[1] * alpha, b = b * (1.0f - alpha) + P.bg[2] * alpha;
        }
        dst[3 * i] = r, dst[3 * i + 1] = g, dst[3 * i + 2] = b;
    }
    return 0;
}

// aux_buffer_kernel! (src/film.jl:435-483); outputs in Julia [h,w] layout
int32_t hko_fill_aux(hko_scene* s, const hk_camera* cam, int32_t w, int32_t h, int32_t has_infinite_lights, float* albedo, float* normal, float* depth) {
    Scene& sc = s->sc;
    const float miss_depth = has_infinite_lights ? 1e30f : INF_F;
    const long n = (long)h * w;
#pragma omp parallel for schedule(dynamic, 256)
    for (long i = 0; i < n; ++i) {
        int row = (int)(i % h) + 1, col = (int)(i / h) + 1;
        CamRay cr = generate_ray(*cam, V2(((float)col - 1.0f) + 0.5f, ((float)row - 1.0f) + 0.5f), V2(0.5f, 0.5f), 0.0f);
        Hit hit = sc.accel.closest_hit(cr.o, cr.d, INF_F);
        float alb = 0.0f, d = miss_depth;
        V3 nn(0.0f);
        if (hit.prim >= 0) {
            V3 v0, v1, v2;
            tri_vertices(sc, hit.prim, v0, v1, v2);
            nn = normalize(cross(v1 - v0, v2 - v0));
            V3 dd = (cr.o + cr.d * hit.t) - cr.o;
            d = std::sqrt(dd.x * dd.x + dd.y * dd.y + dd.z * dd.z);
            alb = 0.8f;
        }
        albedo[3 * i] = albedo[3 * i + 1] = albedo[3 * i + 2] = alb;
        normal[3 * i] = nn.x, normal[3 * i + 1] = nn.y, normal[3 * i + 2] = nn.z;
        depth[i] = d;
    }
    return 0;
}

// OpenMP team size: tiny test frames run faster on a few threads than on a 256-core host
void hko_set_threads(int32_t n) {
#if defined(_OPENMP)
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}
int32_t hko_max_threads(void) {
#if defined(_OPENMP)
    return omp_get_max_threads();
#else
    return 1;
#endif
}

// ---- point-wise helpers for the independent float64 pins (tests/ref64.py, tests/test_independent_pins.py) ----
// Trowbridge-Reitz: out[8] per point = D(wm), Lambda(w), G1(w), G(w, wm taken as wi), pdf(w, wm), sample_wm(w, u).xyz
void hko_tr(int32_t n, const float* w3, const float* wm3, const float* u2, float ax, float ay, float* out) {
    for (int i = 0; i < n; ++i) {
        V3 w(w3[3 * i], w3[3 * i + 1], w3[3 * i + 2]), wm(wm3[3 * i], wm3[3 * i + 1], wm3[3 * i + 2]);
        float* r = out + 8 * (size_t)i;
        r[0] = tr_d(wm, ax, ay);
        r[1] = tr_lambda(w, ax, ay);
        r[2] = tr_g1(w, ax, ay);
        r[3] = tr_g(w, wm, ax, ay);
        r[4] = tr_pdf(w, wm, ax, ay);
        V3 s = tr_sample_wm(w, V2(u2[2 * i], u2[2 * i + 1]), ax, ay);
        r[5] = s.x, r[6] = s.y, r[7] = s.z;
    }
}
// Henyey-Greenstein: out[5] = sample_hg(g, wo, u) -> wi.xyz, pdf; hg_p(g, cos_in)
void hko_hg(int32_t n, float g, const float* wo3, const float* u2, const float* cos_in, float* out) {
    for (int i = 0; i < n; ++i) {
        float pdf;
        V3 wi = sample_hg(g, V3(wo3[3 * i], wo3[3 * i + 1], wo3[3 * i + 2]), V2(u2[2 * i], u2[2 * i + 1]), pdf);
        float* r = out + 5 * (size_t)i;
        r[0] = wi.x, r[1] = wi.y, r[2] = wi.z, r[3] = pdf, r[4] = hg_p(g, cos_in[i]);
    }
}
// equal-area mapping: out[5] = square_to_sphere(uv).xyz, sphere_to_square(dir).uv
void hko_equal_area(int32_t n, const float* uv2, const float* dir3, float* out) {
    for (int i = 0; i < n; ++i) {
        V3 d = equal_area_square_to_sphere(V2(uv2[2 * i], uv2[2 * i + 1]));
        V2 q = equal_area_sphere_to_square(V3(dir3[3 * i], dir3[3 * i + 1], dir3[3 * i + 2]));
        float* r = out + 5 * (size_t)i;
        r[0] = d.x, r[1] = d.y, r[2] = d.z, r[3] = q.x, r[4] = q.y;
    }
}
// Distribution2D of an hk_envmap record: out[5] = sample_continuous(u) -> uv.xy, pdf; pdf(uv_in)
void hko_dist2d(const hk_envmap* e, int32_t n, const float* u2, const float* uv_in2, float* out) {
    for (int i = 0; i < n; ++i) {
        float pdf;
        V2 s = dist2d_sample(*e, V2(u2[2 * i], u2[2 * i + 1]), pdf);
        float* r = out + 4 * (size_t)i;
        r[0] = s.x, r[1] = s.y, r[2] = pdf, r[3] = dist2d_pdf(*e, V2(uv_in2[2 * i], uv_in2[2 * i + 1]));
    }
}
// node_importance of light-BVH node `node_idx` (0-based) of a scene at n points
void hko_node_importance(hko_scene* s, int32_t node_idx, int32_t n, const float* p3, const float* n3, float* out) {
    const LightBVHNode& nd = s->sc.sampler.nodes[node_idx];
    for (int i = 0; i < n; ++i) out[i] = node_importance(nd, V3(p3[3 * i], p3[3 * i + 1], p3[3 * i + 2]), V3(n3[3 * i], n3[3 * i + 1], n3[3 * i + 2]));
}
// cosine_sample_hemisphere / concentric disk (sampler/sampling.jl)
void hko_cosine_hemisphere(int32_t n, const float* u2, float* out3) {
    for (int i = 0; i < n; ++i) {
        V3 d = cosine_sample_hemisphere(V2(u2[2 * i], u2[2 * i + 1]));
        out3[3 * i] = d.x, out3[3 * i + 1] = d.y, out3[3 * i + 2] = d.z;
    }
}

// known-answer helpers
uint64_t hko_murmur64a(const uint8_t* data, int32_t n, uint64_t seed) { return murmur_hash_64a(data, n, seed); }
uint64_t hko_mix_bits(uint64_t v) { return mix_bits(v); }
void hko_pcg32(uint64_t seq, uint64_t seed, int32_t has_seed, int32_t n, uint32_t* out_u32, float* out_f32) {
    PCG32 r = has_seed ? pcg32_init(seq, seed) : pcg32_init(seq);
    PCG32 r2 = r;
    for (int i = 0; i < n; ++i) {
        out_u32[i] = pcg32_uniform_u32(r);
        out_f32[i] = pcg32_uniform_f32(r2);
    }
}
float hko_fresnel_dielectric(float c, float eta) { return fresnel_dielectric(c, eta); }
float hko_fr_complex(float c, float eta, float k) { return fr_complex(c, eta, k); }
float hko_filter_eval(const hk_integrator_params* ip, float x, float y) { return filter_evaluate(make_filter_params(*ip), x, y); }
void hko_filter_sample(const hk_integrator_params* ip, int32_t n, const float* u2, float* out3, float* func_integral) {
    FilterParams fp = make_filter_params(*ip);
    FilterSampler fs = build_filter_sampler(fp);
    if (func_integral) *func_integral = fs.valid ? fs.func_integral : 0.0f;
    for (int i = 0; i < n; ++i) {
        FilterSample s = filter_sample(fp, fs, V2(u2[2 * i], u2[2 * i + 1]));
        out3[3 * i] = s.px;
        out3[3 * i + 1] = s.py;
        out3[3 * i + 2] = s.weight;
    }
}
void hko_wavelengths(int32_t n, const float* u, float* out8) {
    for (int i = 0; i < n; ++i) {
        Wavelengths w = sample_wavelengths_visible(u[i]);
        for (int k = 0; k < 4; ++k) {
            out8[8 * i + k] = w.lambda[k];
            out8[8 * i + 4 + k] = w.pdf[k];
        }
    }
}
float hko_sample_d65(float l) { return sample_d65(l); }

}  // extern "C"

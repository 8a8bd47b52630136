// hk_kernels.hip — wavefront VolPath kernels for gfx950 (wave64).
//
// One "pass" carries S samples of every pixel through the bounce loop at once (paths = pixels x S, laid
// out in 8x8 pixel tiles so a wave's camera rays are coherent).  Per bounce the host enqueues, with no
// readback:
//     k_trace     persistent closest-hit traversal (LDS per-lane stacks) -> hit records, paths
//                 ballot-compacted into one queue per material kind (+ escaped queue)            [K3]
//     k_escaped   environment/ambient lights for escaped paths                                   [K7]
//     k_shade<K>  one specialisation per material kind present: emission MIS, light-BVH NEE,
//                 BSDF sample, Russian roulette, next ray; Sobol dims generated in registers     [K2,K8,K9,K11]
//     k_shadow    shadow-segment traversal, adds Ld*T/mis to the path's radiance                 [K10]
// then k_film folds the S samples of each pixel into the film accumulators in sample order        [K12]
// Queue sizes live in device memory (counters[depth][queue]); every kernel sizes itself from them.
#include <hip/hip_runtime.h>

#include "hikari_mi355x.h"
#include "hk_device.h"

using namespace hkd;

namespace {

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }

// Wave-private segmented queues: wave w owns entries [w*wave_cap, (w+1)*wave_cap) of every queue and the
// count word counts[..][w].  A path never leaves the wave that generated its camera ray, so compaction is pure
// ballot/popcount arithmetic: no atomics, no cursors, deterministic order.  (A single global queue counter
// serialises at ~88 atomics/us on this chip: 40 k wave-pushes per launch cost ~0.45 ms per kernel.)
struct WaveQ {
    uint32_t* base;
    int count;  // wave-uniform
};
__device__ __forceinline__ int global_wave() { return (int)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)); }
__device__ __forceinline__ int physical_waves() { return (int)(gridDim.x * (blockDim.x >> 6)); }
// A kernel is launched with as many physical waves as are resident for ITS register/LDS budget; each physical
// wave walks the virtual wave segments v = p, p + P, p + 2P, ... so all kernels share the same W segments.
#define HK_FOR_EACH_WAVE_SEGMENT(gw, st) for (int gw = global_wave(); gw < (st).n_waves; gw += physical_waves())
__device__ __forceinline__ WaveQ wq_open(uint32_t* q, const DPathState& st, int gw) { return WaveQ{q + (size_t)gw * st.wave_cap, 0}; }
__device__ __forceinline__ void wq_push(WaveQ& q, uint32_t value, bool active) {
    unsigned long long mask = __ballot(active);
    if (active) q.base[q.count + __popcll(mask & ((1ull << lane_id()) - 1ull))] = value;
    q.count += __popcll(mask);
}
__device__ __forceinline__ int* count_ptr(const DPathState& st, int depth, int q, int gw) { return st.counters + ((size_t)(depth * Q_COUNT + q) * st.n_waves + gw); }
__device__ __forceinline__ void wq_close(const WaveQ& q, int* cnt) {
    if (lane_id() == 0) *cnt = q.count;
}

// per-wave statistics rows (summed on the host): plain read-modify-write, the row belongs to this wave
__device__ __forceinline__ void wave_add(unsigned long long* dst, unsigned v) {
    unsigned s = v;
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off);
    if (lane_id() == 0 && s) *dst += (unsigned long long)s;
}

__device__ __forceinline__ void slot_to_pixel(const DFrame& fr, int slot_in_sample, int& px, int& py, bool& inside) {
    int tile = slot_in_sample >> 6, l = slot_in_sample & 63;
    int tx = tile % fr.tiles_x, ty = tile / fr.tiles_x;
    px = tx * 8 + (l & 7);
    py = ty * 8 + (l >> 3);
    inside = px < fr.width && py < fr.height;
}

}  // namespace

// ---------------------------------------------------------------------------------------------------
// K1: camera rays (volpath.jl:125-205).  One thread per path slot of the pass.
// ---------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_camera(DPathState st, DFrame fr, DTables T, DFilter flt, DCamera cam, DSobol sob, int initial_medium) {
    const int total = fr.n_pixels_padded * fr.samples_in_pass;
    const int n_chunks = total >> 6;
    HK_FOR_EACH_WAVE_SEGMENT(gw, st) {
    WaveQ out = wq_open(st.ray_q[0], st, gw);
    // wave w generates chunks w, w+W, w+2W, ... (8x8 pixel tiles interleaved across waves for load balance)
    for (int chunk = gw; chunk < n_chunks; chunk += st.n_waves) {
        int slot = chunk * 64 + lane_id();
        int k = slot / fr.n_pixels_padded;
        int px, py;
        bool active;
        slot_to_pixel(fr, slot - k * fr.n_pixels_padded, px, py, active);
        if (active) {
            int sample_idx = fr.first_sample + k * fr.sample_stride;
            int x = px + 1, y = py + 1;  // 1-based pixel coordinates (Q1)
            SobolCtx sc = sobol_ctx(sob, T.sobol, x, y, sample_idx);
            float wavelength_u = sobol_1d(sc, 1);
            v2 jit = sobol_2d(sc, 3);
            // dims 4 (time) and 6 (lens) only matter with a finite aperture: ray.time is carried by the reference
            // but never read on this path (no motion blur in VolPath), and the lens sample is unused when
            // lens_radius == 0 (perspective.jl:103-112).  Both draws are pure functions, so skipping them is exact.
            float time_u = 0.0f;
            v2 lens = mk2(0.0f, 0.0f);
            if (cam.lens_radius > 0) {
                time_u = sobol_1d(sc, 4);
                lens = sobol_2d(sc, 6);
            }
            float fx, fy, fw;
            filter_sample(flt, jit, fx, fy, fw);
            S4 lambda, pdf;
            sample_wavelengths_visible(wavelength_u, lambda, pdf);
            v2 pfilm = mk2((float)x + 0.5f + fx, (float)fr.height - (float)y + 1.0f + 0.5f + fy);  // Q2
            v3 ro, rd;
            float time;
            generate_ray(cam, pfilm, lens, time_u, ro, rd, time);
            st.ray_o[slot] = make_float4(ro.x, ro.y, ro.z, INF_F);
            st.ray_d[slot] = make_float4(rd.x, rd.y, rd.z, 0.0f);
            st4(&st.lambda[slot], lambda);
            st4(&st.pdf[slot], pdf);
            st4(&st.beta[slot], s4(1.0f));
            st4(&st.r_u[slot], s4(1.0f));
            st4(&st.r_l[slot], s4(1.0f));
            st4(&st.L[slot], s4(0.0f));
            st.filter_w[slot] = fw;
            st.flags[slot] = (uint32_t)(*st.initial_medium + 1) << 16;
        }
        wq_push(out, (uint32_t)slot, active);
    }
    wq_close(out, count_ptr(st, 0, Q_RAY, gw));
    }
}

// ---------------------------------------------------------------------------------------------------
// K3: closest-hit traversal + classification (intersection.jl:188-269).
// ---------------------------------------------------------------------------------------------------
template <bool COUNT>
__global__ void __launch_bounds__(HK_TRACE_BLOCK) k_trace(DPathState st, DScene sc, DTables T, DFrame fr, int depth, DStats* stats) {
    __shared__ int lds_stack[(HK_TRACE_BLOCK / 64) * HK_LDS_STACK * 64];
    int* stack = lds_stack + (threadIdx.x >> 6) * (HK_LDS_STACK * 64);
    const int lane = lane_id();
    unsigned n_nodes = 0, n_tris = 0, n_casts = 0, n_hits = 0;
    HK_FOR_EACH_WAVE_SEGMENT(gw, st) {
    const uint32_t* __restrict__ queue = st.ray_q[depth & 1] + (size_t)gw * st.wave_cap;
    const int n = *count_ptr(st, depth, Q_RAY, gw);
    WaveQ q_escaped = wq_open(st.escaped_q, st, gw);
    WaveQ q_medium = wq_open(st.medium_q, st, gw);
    int kind_count[HK_MAX_KINDS];
#pragma unroll
    for (int k = 0; k < HK_MAX_KINDS; ++k) kind_count[k] = 0;
    for (int base = 0; base < n; base += 64) {
        int i = base + lane;
        bool active = i < n;
        uint32_t slot = active ? queue[i] : 0u;
        int kind = -1;  // -1 none, -2 escaped, >= 0 material kind
        bool in_medium = false;
        if (active && sc.n_media > 0 && (st.flags[slot] >> 16) != 0u) {
            // ray travels inside a medium: one cast (no alpha test, intersection.jl:198-221), then delta tracking
            // (k_medium) decides whether the stored surface hit is ever reached
            in_medium = true;
            float4 O = st.ray_o[slot], D = st.ray_d[slot];
            bool dummy;
            ++n_casts;
            HitRec h = traverse<0, COUNT>(sc, mk3(O.x, O.y, O.z), mk3(D.x, D.y, D.z), O.w, stack, lane, n_nodes, n_tris, dummy);
            if (h.prim >= 0) {
                ++n_hits;
                st.hit[slot] = make_float4(h.t, __int_as_float(h.prim), h.u, h.v);
                st.mat_id[slot] = sc.mis[sc.meta[h.prim].mi].material;
            } else
                st.hit[slot] = make_float4(INF_F, __int_as_float(-1), 0.0f, 0.0f);
        }
        wq_push(q_medium, slot, in_medium);
        if (active && !in_medium) {
            float4 O = st.ray_o[slot], D = st.ray_d[slot];
            v3 ro = mk3(O.x, O.y, O.z), rd = mk3(D.x, D.y, D.z);
            float tmax = O.w;
            // alpha-test loop: alpha-killed surfaces are skipped without consuming depth (<= 16 casts)
            for (int it = 0; it < 16; ++it) {
                bool dummy;
                ++n_casts;
                HitRec h = traverse<0, COUNT>(sc, ro, rd, it == 0 ? tmax : INF_F, stack, lane, n_nodes, n_tris, dummy);
                if (h.prim < 0) {
                    kind = -2;
                    break;
                }
                ++n_hits;
                DTriMeta meta = sc.meta[h.prim];
                int mat = sc.mis[meta.mi].material;
                if (!sc.all_opaque) {
                    float w = 1.0f - h.u - h.v;
                    v2 uv = uv_at(sc, h.prim, w, h.u, h.v);
                    float alpha = surface_alpha(sc, mat, uv);
                    if (alpha < 1.0f) {
                        PCG32 rng = pcg32_init(pbrt_hash(ro), pbrt_hash(rd));
                        if (pcg32_f32(rng) > alpha) {
                            v3 pi = ro + rd * h.t;
                            v3 ng = geometric_normal(sc, h.prim);
                            v3 off = dot(rd, ng) > 0.0f ? ng : -ng;
                            ro = pi + off * 1e-4f;
                            continue;
                        }
                    }
                }
                // MixMaterial is resolved here so the queue is sorted by the *final* material kind
                if (sc.materials[mat].kind == HK_MAT_MIX) {
                    float w = 1.0f - h.u - h.v;
                    v2 uv = uv_at(sc, h.prim, w, h.u, h.v);
                    mat = resolve_mix_material(sc, mat, ro + rd * h.t, -rd, uv);
                }
                kind = sc.materials[mat].kind;
                if (kind == HK_MAT_MIX) kind = HK_MAT_FALLBACK;
                st.hit[slot] = make_float4(h.t, __int_as_float(h.prim), h.u, h.v);
                st.mat_id[slot] = mat;
                if (it > 0) st.ray_o[slot] = make_float4(ro.x, ro.y, ro.z, INF_F);  // origin after alpha skips
                break;
            }
        }
        // ballot-compact into this wave's per-kind queue segments
        wq_push(q_escaped, slot, kind == -2);
        unsigned long long pending = __ballot(kind >= 0);
        while (pending) {
            int src = __ffsll((long long)pending) - 1;
            int k = __shfl(kind, src);
            bool mine = kind == k;
            unsigned long long m = __ballot(mine);
            int cnt = 0;
#pragma unroll
            for (int kk = 0; kk < HK_MAX_KINDS; ++kk) cnt = (kk == k) ? kind_count[kk] : cnt;
            if (mine) st.mat_q[((size_t)k * st.n_waves + gw) * st.wave_cap + cnt + __popcll(m & ((1ull << lane) - 1ull))] = slot;
            int add = __popcll(m);
#pragma unroll
            for (int kk = 0; kk < HK_MAX_KINDS; ++kk) kind_count[kk] += (kk == k) ? add : 0;
            pending &= ~m;
        }
    }
    wq_close(q_escaped, count_ptr(st, depth, Q_ESCAPED, gw));
    wq_close(q_medium, count_ptr(st, depth, Q_MEDIUM, gw));
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < HK_MAX_KINDS; ++k) *count_ptr(st, depth, Q_MAT0 + k, gw) = kind_count[k];
    }
    }
    stats += global_wave();
    wave_add(&stats->rays_closest, n_casts);
    wave_add(&stats->hits, n_hits);
    if (COUNT) {
        wave_add(&stats->nodes, n_nodes);
        wave_add(&stats->tris, n_tris);
    }
}

// ---------------------------------------------------------------------------------------------------
// K4 + K5 + K6 fused: delta tracking with null collisions (delta-tracking.jl:79-453), direct lighting at the
// scattering vertex (medium-scatter.jl:15-138) and phase-function sampling (medium-scatter.jl:148-216).
// Runs after k_trace: paths that survive to their stored surface hit are appended to the material-kind
// queues (Mix resolved here), paths that leave the scene to the escaped queue.
// ---------------------------------------------------------------------------------------------------
template <bool COUNT>
__global__ void __launch_bounds__(256) k_medium(DPathState st, DScene sc, DTables T, DFrame fr, DSobol sob, int depth, DStats* stats) {
    const int lane = lane_id();
    unsigned n_coll = 0, n_lnodes = 0;
    HK_FOR_EACH_WAVE_SEGMENT(gw, st) {
        const uint32_t* __restrict__ queue = st.medium_q + (size_t)gw * st.wave_cap;
        const int n = *count_ptr(st, depth, Q_MEDIUM, gw);
        WaveQ q_escaped = wq_open(st.escaped_q, st, gw);
        q_escaped.count = *count_ptr(st, depth, Q_ESCAPED, gw);
        WaveQ q_shadow = wq_open(st.shadow_q, st, gw);            // first writer of this depth's shadow / next-ray segments
        WaveQ q_next = wq_open(st.ray_q[(depth + 1) & 1], st, gw);
        int kind_count[HK_MAX_KINDS];
#pragma unroll
        for (int k = 0; k < HK_MAX_KINDS; ++k) kind_count[k] = *count_ptr(st, depth, Q_MAT0 + k, gw);
        for (int base = 0; base < n; base += 64) {
            int i = base + lane;
            bool active = i < n;
            uint32_t slot = active ? queue[i] : 0u;
            int kind = -1;  // -1 terminated, -2 escaped, >= 0 reached its surface hit of that material kind
            bool push_shadow = false, push_ray = false;
            if (active) {
                float4 O = st.ray_o[slot], D = st.ray_d[slot], H = st.hit[slot];
                v3 ro = mk3(O.x, O.y, O.z), rd = mk3(D.x, D.y, D.z);
                const float t_max = H.x;
                const int prim = __float_as_int(H.y);
                uint32_t fl = st.flags[slot];
                const int medium_idx = (int)(fl >> 16) - 1;
                const DMedium& med = sc.media[medium_idx];
                S4 lambda = ld4(&st.lambda[slot]);
                S4 beta = ld4(&st.beta[slot]), r_u = ld4(&st.r_u[slot]), r_l = ld4(&st.r_l[slot]);
                S4 base_a = eval_scaled(med.sigma_a, lambda), base_s = eval_scaled(med.sigma_s, lambda), base_Le = eval_scaled(med.Le, lambda);
                uint64_t rng = lcg_init(ro, rd, t_max);
                MajorantIter it = create_majorant_iterator(med, ro, rd, t_max, lambda);
                bool done = false, scattered = false;
                v3 sp = mk3(0, 0, 0);
                float sg = 0.0f;
                for (int segi = 0; segi < 256 && !done; ++segi) {
                    float seg0, seg1;
                    S4 sm;
                    if (!majorant_next(it, seg0, seg1, sm)) break;
                    float sm0 = sm.x;
                    if (sm0 < 1e-10f) continue;
                    float t = seg0;
                    v3 cur_o = ro + rd * t;
                    for (int k = 0; k < 1024; ++k) {
                        float u = lcg_next(rng);
                        float dt = -logf(maxf(1e-10f, 1.0f - u)) / sm0;
                        float ts = t + dt;
                        if (ts >= seg1) {
                            float dr = seg1 - t;
                            S4 Tm = s4exp((-dr) * sm);
                            float T0 = Tm.x;
                            if (T0 > 1e-10f) {
                                beta = beta * Tm / T0;
                                r_u = r_u * Tm / T0;
                                r_l = r_l * Tm / T0;
                            }
                            break;
                        }
                        S4 Tm = s4exp((-dt) * sm);
                        v3 p = cur_o + rd * dt;
                        ++n_coll;
                        MediumProps mp = sample_point(T, lambda, med, base_a, base_s, base_Le, p);
                        if (!is_black(mp.Le) && depth < fr.max_depth) {
                            float pr = sm0 * Tm.x;
                            if (pr > 1e-10f) {
                                S4 r_e = r_u * sm * Tm / pr;
                                if (!is_black(r_e)) st4(&st.L[slot], ld4(&st.L[slot]) + beta * mp.sigma_a * Tm * mp.Le / (pr * average(r_e)));
                            }
                        }
                        float p_absorb = mp.sigma_a.x / sm0, p_scatter = mp.sigma_s.x / sm0;
                        float ue = lcg_next(rng);
                        if (ue < p_absorb) {
                            done = true;
                            break;
                        } else if (ue < p_absorb + p_scatter) {
                            done = true;
                            if (depth >= fr.max_depth) break;
                            float pdf = Tm.x * mp.sigma_s.x;
                            if (pdf > 1e-10f) {
                                beta = beta * Tm * mp.sigma_s / pdf;
                                r_u = r_u * Tm * mp.sigma_s / pdf;
                            }
                            scattered = true;
                            sp = p;
                            sg = mp.g;
                            break;
                        } else {
                            S4 sn = s4max0(sm - mp.sigma_a - mp.sigma_s);
                            float pdf = Tm.x * sn.x;
                            if (pdf > 1e-10f) {
                                beta = beta * Tm * sn / pdf;
                                r_u = r_u * Tm * sn / pdf;
                                r_l = r_l * Tm * sm / pdf;
                            } else {
                                done = true;
                                break;
                            }
                            t = ts;
                            cur_o = p;
                            if (is_black(beta) || is_black(r_u)) {
                                done = true;
                                break;
                            }
                        }
                    }
                }
                if (scattered) {
                    v3 wo = -rd;
                    int k = (int)slot / fr.n_pixels_padded;
                    int px, py;
                    bool inside;
                    slot_to_pixel(fr, (int)slot - k * fr.n_pixels_padded, px, py, inside);
                    SobolCtx sctx = sobol_ctx(sob, T.sobol, px + 1, py + 1, fr.first_sample + k * fr.sample_stride);
                    const int base_dim = 6 + 7 * depth;
                    // ---- K5: light-BVH NEE with n = 0, HG evaluated with cos = wo.wi (medium-scatter.jl:46-49) ----
                    if (sc.n_lights > 0) {
                        float light_select = sobol_1d(sctx, base_dim + 1);
                        float light_pmf;
                        int light_idx = bvh_sample_light(sc, sp, mk3(0, 0, 0), light_select, light_pmf, n_lnodes);
                        if (light_idx >= 1 && light_idx <= sc.n_lights && light_pmf > 0.0f) {
                            const DLight& sel = sc.lights[light_idx - 1];
                            v2 u_light = mk2(0.0f, 0.0f);
                            if (sel.kind >= HK_LIGHT_AMBIENT) u_light = sobol_2d(sctx, base_dim + 3);
                            LightSample ls = sample_light(sc, T, sel, sp, lambda, u_light);
                            if (ls.pdf > 0.0f && !is_black(ls.Li)) {
                                float phase_val = hg_p(sg, dot(wo, ls.wi));
                                if (phase_val > 0.0f) {
                                    float light_pdf = ls.pdf * light_pmf;
                                    float phase_pdf = ls.is_delta ? 0.0f : phase_val;
                                    float tmx = ls.is_delta ? norm(ls.p_light - sp) - 0.001f : 1.0e6f;
                                    st.sh_o[slot] = make_float4(sp.x, sp.y, sp.z, tmx);
                                    st.sh_d[slot] = make_float4(ls.wi.x, ls.wi.y, ls.wi.z, __int_as_float(medium_idx));
                                    st4(&st.sh_Ld[slot], beta * phase_val * ls.Li);
                                    st4(&st.sh_ru[slot], r_u * phase_pdf);
                                    st4(&st.sh_rl[slot], r_u * light_pdf);
                                    push_shadow = true;
                                }
                            }
                        }
                    }
                    // ---- K6: sample the phase function, continue in the same medium ----
                    int new_depth = depth + 1;
                    if (new_depth < fr.max_depth) {
                        v2 u = sobol_2d(sctx, base_dim + 6);
                        float ppdf;
                        v3 wi = sample_hg(sg, wo, u, ppdf);
                        if (ppdf > 0.0f) {
                            st.ray_o[slot] = make_float4(sp.x, sp.y, sp.z, INF_F);
                            st.ray_d[slot] = make_float4(wi.x, wi.y, wi.z, D.w);
                            st4(&st.beta[slot], beta);
                            st4(&st.r_u[slot], r_u);
                            st4(&st.r_l[slot], r_u / ppdf);
                            st.flags[slot] = (uint32_t)new_depth | (1u << 9) | ((uint32_t)(medium_idx + 1) << 16);  // specular = false, any_non_specular = true
                            push_ray = true;
                        }
                    }
                } else if (!done && !(is_black(beta) || is_black(r_u) || depth >= fr.max_depth)) {
                    // survived to t_max: hand the stored surface hit / the escape over with the updated throughput
                    st4(&st.beta[slot], beta);
                    st4(&st.r_u[slot], r_u);
                    st4(&st.r_l[slot], r_l);
                    if (prim < 0)
                        kind = -2;
                    else {
                        int mat = st.mat_id[slot];
                        if (sc.materials[mat].kind == HK_MAT_MIX) {
                            float w = 1.0f - H.z - H.w;
                            mat = resolve_mix_material(sc, mat, ro + rd * t_max, -rd, uv_at(sc, prim, w, H.z, H.w));
                            st.mat_id[slot] = mat;
                        }
                        kind = sc.materials[mat].kind;
                        if (kind == HK_MAT_MIX) kind = HK_MAT_FALLBACK;
                    }
                }
            }
            wq_push(q_shadow, slot, push_shadow);
            wq_push(q_next, slot, push_ray);
            wq_push(q_escaped, slot, kind == -2);
            unsigned long long pending = __ballot(kind >= 0);
            while (pending) {
                int src = __ffsll((long long)pending) - 1;
                int k = __shfl(kind, src);
                bool mine = kind == k;
                unsigned long long m = __ballot(mine);
                int cnt = 0;
#pragma unroll
                for (int kk = 0; kk < HK_MAX_KINDS; ++kk) cnt = (kk == k) ? kind_count[kk] : cnt;
                if (mine) st.mat_q[((size_t)k * st.n_waves + gw) * st.wave_cap + cnt + __popcll(m & ((1ull << lane) - 1ull))] = slot;
                int add = __popcll(m);
#pragma unroll
                for (int kk = 0; kk < HK_MAX_KINDS; ++kk) kind_count[kk] += (kk == k) ? add : 0;
                pending &= ~m;
            }
        }
        wq_close(q_shadow, count_ptr(st, depth, Q_SHADOW, gw));
        wq_close(q_next, count_ptr(st, depth + 1, Q_RAY, gw));
        wq_close(q_escaped, count_ptr(st, depth, Q_ESCAPED, gw));
        if (lane == 0) {
#pragma unroll
            for (int k = 0; k < HK_MAX_KINDS; ++k) *count_ptr(st, depth, Q_MAT0 + k, gw) = kind_count[k];
        }
    }
    stats += global_wave();
    wave_add(&stats->collisions, n_coll);
    wave_add(&stats->light_nodes, n_lnodes);
}

// K14: which medium is the camera in?  (intersection.jl:690-747)  One lane, result stays on the device.
__global__ void __launch_bounds__(64) k_detect_camera_medium(DPathState st, DScene sc, float cx, float cy, float cz, DStats* stats) {
    __shared__ int lds_stack[HK_LDS_STACK * 64];
    if (threadIdx.x != 0) return;
    v3 d = mk3(0.57735027f, 0.57735027f, 0.57735027f);
    v3 o = mk3(cx, cy, cz);
    int result = -1;
    unsigned a = 0, b = 0, casts = 0;
    for (int it = 0; it < 16; ++it) {
        bool dummy;
        ++casts;
        HitRec h = traverse<0, false>(sc, o, d, INF_F, lds_stack, 0, a, b, dummy);
        if (h.prim < 0) break;
        DMediumInterface mi = sc.mis[sc.meta[h.prim].mi];
        v3 n = geometric_normal(sc, h.prim);
        if (mi.inside != mi.outside) {
            result = dot(-d, n) > 0.0f ? mi.outside : mi.inside;
            break;
        }
        v3 off = dot(d, n) > 0.0f ? n : -n;
        o = (o + d * h.t) + off * 1e-4f;
    }
    *st.initial_medium = result;
    stats->rays_closest += casts;
}

// ---------------------------------------------------------------------------------------------------
// K7: escaped rays (intersection.jl:622-678; lights.jl:408-467).  MIS uses 1/num_lights (Q6).
// ---------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_escaped(DPathState st, DScene sc, DTables T, int depth) {
    HK_FOR_EACH_WAVE_SEGMENT(gw, st) {
    const uint32_t* __restrict__ queue = st.escaped_q + (size_t)gw * st.wave_cap;
    const int n = *count_ptr(st, depth, Q_ESCAPED, gw);
    for (int i = lane_id(); i < n; i += 64) {
        uint32_t slot = queue[i];
        S4 lambda = ld4(&st.lambda[slot]);
        S4 Le = s4(0.0f);
        v3 rd = mk3(0, 0, 1);
        bool have_dir = false;
        for (int li = 0; li < sc.n_lights; ++li) {
            const DLight& l = sc.lights[li];
            if (l.kind == HK_LIGHT_AMBIENT) Le = Le + l.scale * light_spectrum(l, lambda);
            if (l.kind == HK_LIGHT_ENVIRONMENT) {  // bilinear env(dir) * scale, illuminant uplift (lights.jl:408-419)
                if (!have_dir) {
                    float4 D = st.ray_d[slot];
                    rd = mk3(D.x, D.y, D.z);
                    have_dir = true;
                }
                float4 t = env_eval(sc.envmaps[l.Le_tex], rd);
                Le = Le + eval_illuminant(coef_illuminant(T, t.x * l.Le_rgba[0], t.y * l.Le_rgba[1], t.z * l.Le_rgba[2]), lambda);
            }
        }
        S4 beta = ld4(&st.beta[slot]);
        S4 contribution = beta * Le;
        if (is_black(contribution)) continue;
        uint32_t fl = st.flags[slot];
        int pdepth = (int)(fl & 0xff);
        bool specular = (fl >> 8) & 1u;
        S4 r_u = ld4(&st.r_u[slot]);
        S4 fin;
        if (pdepth == 0 || specular)
            fin = contribution / average(r_u);
        else {
            float choice = sc.n_lights > 0 ? 1.0f / (float)sc.n_lights : 0.0f;
            float light_pdf = 0.0f;  // only EnvironmentLight has a pdf (lights.jl:445-467)
            if (sc.n_envmaps > 0)
                for (int li = 0; li < sc.n_lights; ++li) {
                    const DLight& l = sc.lights[li];
                    light_pdf = light_pdf + (l.kind == HK_LIGHT_ENVIRONMENT ? env_pdf_li(sc.envmaps[l.Le_tex], rd) : 0.0f);
                }
            S4 rl = ld4(&st.r_l[slot]) * choice * light_pdf;
            float den = average(r_u + rl);
            fin = den > 1e-10f ? contribution / den : contribution / average(r_u);
        }
        st4(&st.L[slot], ld4(&st.L[slot]) + fin);
    }
    }
}

// ---------------------------------------------------------------------------------------------------
// K8 + K9 + K11 fused per material kind (surface-eval.jl:147-220, 250-341, 396-512).
// ---------------------------------------------------------------------------------------------------
// Register budget: 512 VGPRs per SIMD lane => 3 waves/SIMD need <= 168, 4 need <= 128.  The simple kinds sit a few registers
// above 168 without a hint; asking for 3 waves costs a handful of scratch spills and buys a third more latency hiding.
// The walk kinds (coated diffuse / transmission) are far above: they keep the default.
#ifndef HK_SHADE_WAVES
#define HK_SHADE_WAVES 3
#endif
template <int KIND>
struct ShadeWaves {
    static constexpr int value = (KIND == HK_MAT_COATED_DIFFUSE || KIND == HK_MAT_COATED_DIFFUSE_TRANSMISSION) ? 1 : HK_SHADE_WAVES;
};
#ifndef HK_SHADE_MIN_WAVES
#define HK_SHADE_MIN_WAVES 1
#endif
template <int KIND>
__global__ void __attribute__((amdgpu_flat_work_group_size(256, 256), amdgpu_waves_per_eu(ShadeWaves<KIND>::value))) k_shade(DPathState st, DScene sc, DTables T, DFrame fr, DSobol sob, int depth, int first_kind, DStats* stats) {
    const int lane = lane_id();
    unsigned n_vertices = 0, n_lnodes = 0;
    HK_FOR_EACH_WAVE_SEGMENT(gw, st) {
    const uint32_t* __restrict__ queue = st.mat_q + ((size_t)KIND * st.n_waves + gw) * st.wave_cap;
    const int n = *count_ptr(st, depth, Q_MAT0 + KIND, gw);
    // several kinds append to the same shadow / next-ray segments: continue from the counts left by the kinds before
    WaveQ q_shadow = wq_open(st.shadow_q, st, gw);
    WaveQ q_next = wq_open(st.ray_q[(depth + 1) & 1], st, gw);
    if (!first_kind) {
        q_shadow.count = *count_ptr(st, depth, Q_SHADOW, gw);
        q_next.count = *count_ptr(st, depth + 1, Q_RAY, gw);
    }
    for (int base = 0; base < n; base += 64) {
        int i = base + lane;
        bool active = i < n;
        uint32_t slot = active ? queue[i] : 0u;
        bool push_shadow = false, push_ray = false;
        if (active) {
            ++n_vertices;
            float4 H = st.hit[slot];
            float4 O = st.ray_o[slot], D = st.ray_d[slot];
            v3 ro = mk3(O.x, O.y, O.z), rd = mk3(D.x, D.y, D.z);
            float t_hit = H.x;
            int prim = __float_as_int(H.y);
            Surface sf = surface_at(sc, prim, H.z, H.w, ro, rd, t_hit);
            v3 wo = -rd;
            DTriMeta meta = sc.meta[prim];
            DMediumInterface mi = sc.mis[meta.mi];
            const DMaterial& mat = sc.materials[st.mat_id[slot]];
            S4 lambda = ld4(&st.lambda[slot]);
            S4 beta = ld4(&st.beta[slot]);
            S4 r_u = ld4(&st.r_u[slot]);
            S4 r_l = ld4(&st.r_l[slot]);
            uint32_t fl = st.flags[slot];
            int pdepth = (int)(fl & 0xff);
            bool specular_bounce = (fl >> 8) & 1u, any_non_specular = (fl >> 9) & 1u;
            int medium = (int)(fl >> 16) - 1;

            // ---- K8: emission from an area light hit, MIS against the light-BVH pmf ----
            if (meta.arealight > 0) {
                const DLight& light = sc.lights[meta.arealight - 1];
                S4 Le = arealight_Le(sc, T, light, wo, sf.n, sf.uv, lambda);
                if (!is_black(Le)) {
                    S4 contribution = beta * Le;
                    S4 fin;
                    if (pdepth == 0 || specular_bounce)
                        fin = contribution / average(r_u);
                    else {
                        float choice = bvh_pmf(sc, sf.pi, sf.n, (int)meta.arealight, n_lnodes);
                        float ct = fabsf(dot(sf.n, normalize(rd)));
                        float light_pdf = 0.0f;
                        if (ct > 0.0f && sf.area > 0.0f) light_pdf = choice * ((t_hit * t_hit) / (ct * sf.area));
                        S4 rl = r_l * light_pdf;
                        float den = average(r_u + rl);
                        fin = den > 1e-10f ? contribution / den : contribution / average(r_u);
                    }
                    st4(&st.L[slot], ld4(&st.L[slot]) + fin);
                }
            }

            // pixel coordinates for the Sobol dimensions of this bounce (volpath.jl:252-262, Q19)
            int k = (int)slot / fr.n_pixels_padded;
            int px, py;
            bool inside;
            slot_to_pixel(fr, (int)slot - k * fr.n_pixels_padded, px, py, inside);
            SobolCtx sctx = sobol_ctx(sob, T.sobol, px + 1, py + 1, fr.first_sample + k * fr.sample_stride);
            // every path in the depth-d queues has work.depth == d, so the dimension (and its scramble hashes) is
            // wave-uniform: derived from the kernel argument it stays in scalar registers
            const int base_dim = 6 + 7 * depth;

            // ---- K9: next-event estimation through the light BVH ----
            if (sc.n_lights > 0) {
                float light_select = sobol_1d(sctx, base_dim + 1);
                float light_pmf;
                int light_idx = bvh_sample_light(sc, sf.pi, sf.ns, light_select, light_pmf, n_lnodes);
                if (light_idx >= 1 && light_idx <= sc.n_lights && light_pmf > 0.0f) {
                    const DLight& sel = sc.lights[light_idx - 1];
                    // delta lights ignore the 2-D sample (lights.jl:39-131): draw it only for lights that use it
                    v2 u_light = mk2(0.0f, 0.0f);
                    if (sel.kind >= HK_LIGHT_AMBIENT) u_light = sobol_2d(sctx, base_dim + 3);
                    LightSample ls = sample_light(sc, T, sel, sf.pi, lambda, u_light);
                    if (ls.pdf > 0.0f && !is_black(ls.Li)) {
                        float bsdf_pdf;
                        S4 f = eval_bsdf<KIND>(sc, T, mat, wo, ls.wi, sf.ns, TexCtx(sf.uv, meta.prim_index, H.z, H.w), lambda, bsdf_pdf);
                        if (!is_black(f)) {
                            float ct = fabsf(dot(ls.wi, sf.ns));
                            S4 Ld = beta * f * ls.Li * ct;
                            if (!is_black(Ld)) {
                                v3 off = 1e-4f * sf.ns;  // NB: shading normal (Q9)
                                v3 so = dot(ls.wi, sf.ns) > 0.0f ? sf.pi + off : sf.pi - off;
                                v3 tl = ls.p_light - so;
                                float tmax = sqrtf(dot(tl, tl)) - 1e-3f;
                                float nbp = ls.is_delta ? 0.0f : bsdf_pdf;
                                st.sh_o[slot] = make_float4(so.x, so.y, so.z, tmax);
                                st.sh_d[slot] = make_float4(ls.wi.x, ls.wi.y, ls.wi.z, __int_as_float(medium));
                                st4(&st.sh_Ld[slot], Ld);
                                st4(&st.sh_ru[slot], r_u * nbp);
                                st4(&st.sh_rl[slot], (r_u * ls.pdf) * light_pmf);
                                push_shadow = true;
                            }
                        }
                    }
                }
            }

            // ---- K11: BSDF sampling, throughput, Russian roulette, continuation ray ----
            int new_depth = pdepth + 1;
            if (new_depth < fr.max_depth) {
                // the 1-D component sample is read only by BSDFs that choose a lobe (Glass and the layered kinds)
                float uc = (KIND == HK_MAT_GLASS || KIND > HK_MAT_CONDUCTOR) && KIND != HK_MAT_FALLBACK ? sobol_1d(sctx, base_dim + 4) : 0.0f;
                v2 u = (KIND == HK_MAT_MIRROR || KIND == HK_MAT_GLASS || KIND == HK_MAT_THIN_DIELECTRIC) ? mk2(0.0f, 0.0f) : sobol_2d(sctx, base_dim + 6);
                bool regularize = fr.regularize && any_non_specular;
                BSDFSample s = sample_bsdf<KIND>(sc, T, mat, wo, sf.ns, TexCtx(sf.uv, meta.prim_index, H.z, H.w), lambda, u, uc, regularize);
                if (s.pdf > 0.0f && !is_black(s.f)) {
                    float ct = fabsf(dot(s.wi, sf.ns));
                    S4 nb = s.is_specular ? beta * s.f : beta * s.f * ct / s.pdf;
                    S4 nrl = s.is_specular ? r_u : r_u / s.pdf;
                    bool cont = true;
                    if (new_depth > 3) {  // russian_roulette_spectral, min_depth fixed at 3 (Q7)
                        float rr = sobol_1d(sctx, base_dim + 7);
                        float q = maxf(0.05f, 1.0f - max_component(nb));
                        if (rr < q)
                            cont = false;
                        else
                            nb = nb * (1.0f / (1.0f - q));
                    }
                    if (cont) {
                        int new_medium = (mi.inside != mi.outside) ? (dot(s.wi, sf.n) > 0.0f ? mi.outside : mi.inside) : medium;
                        v3 off = dot(s.wi, sf.n) > 0.0f ? sf.n : -sf.n;
                        v3 no = sf.pi + off * 0.0001f;
                        st.ray_o[slot] = make_float4(no.x, no.y, no.z, INF_F);
                        st.ray_d[slot] = make_float4(s.wi.x, s.wi.y, s.wi.z, 0.0f);  // time = 0 (Q17)
                        st4(&st.beta[slot], nb);
                        st4(&st.r_l[slot], nrl);  // r_u unchanged
                        bool ans = any_non_specular || !s.is_specular;
                        st.flags[slot] = (uint32_t)new_depth | ((s.is_specular ? 1u : 0u) << 8) | ((ans ? 1u : 0u) << 9) | ((uint32_t)(new_medium + 1) << 16);
                        push_ray = true;
                    }
                }
            }
        }
        wq_push(q_shadow, slot, push_shadow);
        wq_push(q_next, slot, push_ray);
    }
    wq_close(q_shadow, count_ptr(st, depth, Q_SHADOW, gw));
    wq_close(q_next, count_ptr(st, depth + 1, Q_RAY, gw));
    }
    stats += global_wave();
    wave_add(&stats->vertices, n_vertices);
    wave_add(&stats->light_nodes, n_lnodes);
}

// ---------------------------------------------------------------------------------------------------
// K10: shadow rays (intersection.jl:302-406, 565-600).  Surface-only scenes: one segment, early exit on
// any opaque hit.  Medium-transition / alpha surfaces are walked through (<= 10 segments).
// ---------------------------------------------------------------------------------------------------
// SURFACES_ONLY (no media, every surface opaque) is the lean instantiation: one any-hit cast, ~1/3 of the registers of the
// general walk (whose ratio tracking + run-time RGB uplift would otherwise set the occupancy of every scene).
template <bool COUNT, bool SURFACES_ONLY>
__global__ void __launch_bounds__(HK_TRACE_BLOCK) k_shadow(DPathState st, DScene sc, DTables T, int depth, DStats* stats) {
    __shared__ int lds_stack[(HK_TRACE_BLOCK / 64) * HK_LDS_STACK * 64];
    int* stack = lds_stack + (threadIdx.x >> 6) * (HK_LDS_STACK * 64);
    const int lane = lane_id();
    unsigned n_nodes = 0, n_tris = 0, n_casts = 0, n_hits = 0, n_coll = 0;
    HK_FOR_EACH_WAVE_SEGMENT(gw, st) {
    const uint32_t* __restrict__ queue = st.shadow_q + (size_t)gw * st.wave_cap;
    const int n = *count_ptr(st, depth, Q_SHADOW, gw);
    for (int base = 0; base < n; base += 64) {
        int i = base + lane;
        if (i >= n) continue;
        uint32_t slot = queue[i];
        float4 O = st.sh_o[slot], D = st.sh_d[slot];
        v3 ro = mk3(O.x, O.y, O.z), dir = mk3(D.x, D.y, D.z);
        float t_remaining = O.w;
        int medium = __float_as_int(D.w);
        S4 T_ray = s4(1.0f), tr_u = s4(1.0f), tr_l = s4(1.0f);
        S4 lambda = s4(0.0f);
        if (sc.n_media > 0) lambda = ld4(&st.lambda[slot]);
        bool visible = false, done = false;
        if (SURFACES_ONLY) {
            if (t_remaining >= 1e-6f) {
                bool opaque;
                ++n_casts;
                HitRec h = traverse<1, COUNT>(sc, ro, dir, t_remaining, stack, lane, n_nodes, n_tris, opaque);
                visible = h.prim < 0;
                if (!visible) ++n_hits;
            }
            done = true;
        }
        for (int seg = 0; seg < 10 && !done; ++seg) {
            if (t_remaining < 1e-6f) break;
            bool opaque;
            ++n_casts;
            HitRec h = traverse<1, COUNT>(sc, ro, dir, t_remaining, stack, lane, n_nodes, n_tris, opaque);
            if (h.prim < 0) {
                if (medium >= 0) {  // transmittance of the remaining distance (intersection.jl:326-336)
                    S4 sT, su, sl;
                    ratio_tracking(T, sc.media[medium], ro, dir, t_remaining, lambda, sT, su, sl, n_coll);
                    T_ray = T_ray * sT;
                    tr_u = tr_u * su;
                    tr_l = tr_l * sl;
                }
                visible = true;
                done = true;
                break;
            }
            ++n_hits;
            if (opaque) {
                done = true;
                break;
            }
            DTriMeta meta = sc.meta[h.prim];
            DMediumInterface mi = sc.mis[meta.mi];
            v3 ng = geometric_normal(sc, h.prim);
            bool entering = dot(dir, ng) < 0.0f;
            if (mi.inside == mi.outside) {
                float w = 1.0f - h.u - h.v;
                float alpha = surface_alpha(sc, mi.material, uv_at(sc, h.prim, w, h.u, h.v));
                bool pass = false;
                if (alpha < 1.0f) {
                    PCG32 rng = pcg32_init(pbrt_hash(ro), pbrt_hash(dir));
                    pass = pcg32_f32(rng) > alpha;
                }
                if (!pass) {
                    done = true;
                    break;
                }
            }
            if (medium >= 0) {  // transmittance up to this surface
                S4 sT, su, sl;
                ratio_tracking(T, sc.media[medium], ro, dir, h.t, lambda, sT, su, sl, n_coll);
                T_ray = T_ray * sT;
                tr_u = tr_u * su;
                tr_l = tr_l * sl;
            }
            if (mi.inside != mi.outside) {
                if (is_black(T_ray)) {
                    visible = true;
                    done = true;
                    break;
                }
                medium = entering ? mi.inside : mi.outside;
            }
            ro = ro + dir * (h.t + 1e-4f);
            t_remaining = t_remaining - h.t - 1e-4f;
        }
        if (visible && !is_black(T_ray)) {
            S4 mis = ld4(&st.sh_ru[slot]) * tr_u + ld4(&st.sh_rl[slot]) * tr_l;
            float den = average(mis);
            if (den > 1e-10f) {
                S4 fin = ld4(&st.sh_Ld[slot]) * T_ray / den;
                if (!is_black(fin)) st4(&st.L[slot], ld4(&st.L[slot]) + fin);
            }
        }
        (void)medium;
    }
    }
    stats += global_wave();
    wave_add(&stats->collisions, n_coll);
    wave_add(&stats->rays_shadow, n_casts);
    wave_add(&stats->hits, n_hits);
    if (COUNT) {
        wave_add(&stats->sh_nodes, n_nodes);
        wave_add(&stats->sh_tris, n_tris);
    }
}

// ---------------------------------------------------------------------------------------------------
// K12: spectral -> RGB, firefly clamp, filter-weighted accumulation (volpath.jl:326-375).  The S samples of
// a pixel are folded in sample order, so the fp32 sums equal the reference's sample-by-sample sums.
// ---------------------------------------------------------------------------------------------------
template <typename ACC>
__global__ void __launch_bounds__(256) k_film(DPathState st, DFrame fr, DTables T, ACC* __restrict__ accum) {
    int n = fr.n_pixels_padded;
    size_t N = (size_t)fr.width * fr.height;
    for (int tid = blockIdx.x * blockDim.x + threadIdx.x; tid < n; tid += gridDim.x * blockDim.x) {
        int px, py;
        bool inside;
        slot_to_pixel(fr, tid, px, py, inside);
        if (!inside) continue;
        size_t p = (size_t)py * fr.width + px;
        ACC r = accum[3 * p], g = accum[3 * p + 1], b = accum[3 * p + 2], w = accum[3 * N + p];
        for (int k = 0; k < fr.samples_in_pass; ++k) {
            size_t slot = (size_t)k * n + tid;
            v3 rgb = spectral_to_rgb_clamped(T, ld4(&st.L[slot]), ld4(&st.lambda[slot]), ld4(&st.pdf[slot]), fr.max_component_value);
            float fw = st.filter_w[slot];
            r += (ACC)(fw * rgb.x);
            g += (ACC)(fw * rgb.y);
            b += (ACC)(fw * rgb.z);
            w += (ACC)fw;
        }
        accum[3 * p] = r;
        accum[3 * p + 1] = g;
        accum[3 * p + 2] = b;
        accum[3 * N + p] = w;
    }
}

// K13 (volpath.jl:384-417): out = Julia Matrix{RGB{Float32}}[height,width] column-major
template <typename ACC>
__global__ void k_finalize(const ACC* __restrict__ accum, float* __restrict__ out, int width, int height) {
    size_t N = (size_t)width * height;
    for (size_t p = blockIdx.x * (size_t)blockDim.x + threadIdx.x; p < N; p += (size_t)gridDim.x * blockDim.x) {
        int px = (int)(p % width), py = (int)(p / width);
        ACC w = accum[3 * N + p];
        float r = 0.0f, g = 0.0f, b = 0.0f;
        if (w > (ACC)0) {
            ACC inv = (ACC)1 / w;
            r = (float)(accum[3 * p] * inv);
            g = (float)(accum[3 * p + 1] * inv);
            b = (float)(accum[3 * p + 2] * inv);
        }
        float* o = out + 3 * ((size_t)py + (size_t)height * px);
        o[0] = r;
        o[1] = g;
        o[2] = b;
    }
}

// ---------------------------------------------------------------------------------------------------
// sub-kernel entry points used by the parity tests
// ---------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(HK_TRACE_BLOCK) k_test_trace(DScene sc, int n, const float* o3, const float* d3, const float* tmax, float* out_t, int* out_prim,
                                                               float* out_uv) {
    __shared__ int lds_stack[(HK_TRACE_BLOCK / 64) * HK_LDS_STACK * 64];
    int* stack = lds_stack + (threadIdx.x >> 6) * (HK_LDS_STACK * 64);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        unsigned a = 0, b = 0;
        bool dummy;
        HitRec h = traverse<0, false>(sc, mk3(o3[3 * i], o3[3 * i + 1], o3[3 * i + 2]), mk3(d3[3 * i], d3[3 * i + 1], d3[3 * i + 2]), tmax[i], stack, lane_id(), a, b,
                                      dummy);
        out_t[i] = h.prim >= 0 ? h.t : INF_F;
        out_prim[i] = h.prim;
        out_uv[2 * i] = h.prim >= 0 ? h.u : 0.0f;
        out_uv[2 * i + 1] = h.prim >= 0 ? h.v : 0.0f;
    }
}
__global__ void k_test_sobol(DTables T, DSobol sob, int n, const int* px, const int* py, const int* sidx, const int* dim, float* o1, float* o2) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        SobolCtx c = sobol_ctx(sob, T.sobol, px[i], py[i], sidx[i]);
        o1[i] = sobol_1d(c, dim[i]);
        v2 v = sobol_2d(c, dim[i]);
        o2[2 * i] = v.x;
        o2[2 * i + 1] = v.y;
    }
}
__global__ void k_test_camera(DTables T, DFilter flt, DCamera cam, DSobol sob, int height, int n, const int* px, const int* py, const int* sidx, float* out15) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        int x = px[i], y = py[i];
        SobolCtx sc = sobol_ctx(sob, T.sobol, x, y, sidx[i]);
        float wu = sobol_1d(sc, 1);
        v2 jit = sobol_2d(sc, 3);
        float tu = sobol_1d(sc, 4);
        v2 lens = sobol_2d(sc, 6);
        float fx, fy, fw;
        filter_sample(flt, jit, fx, fy, fw);
        S4 lambda, pdf;
        sample_wavelengths_visible(wu, lambda, pdf);
        v2 pfilm = mk2((float)x + 0.5f + fx, (float)height - (float)y + 1.0f + 0.5f + fy);
        v3 ro, rd;
        float time;
        generate_ray(cam, pfilm, lens, tu, ro, rd, time);
        float* o = out15 + 15 * (size_t)i;
        o[0] = lambda.x; o[1] = lambda.y; o[2] = lambda.z; o[3] = lambda.w;
        o[4] = pdf.x; o[5] = pdf.y; o[6] = pdf.z; o[7] = pdf.w;
        o[8] = fw;
        o[9] = ro.x; o[10] = ro.y; o[11] = ro.z;
        o[12] = rd.x; o[13] = rd.y; o[14] = rd.z;
    }
}
__global__ void k_test_uplift(DTables T, int mode, int n, const float* rgb, const float* lam, float* out) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        S4 l = s4(lam[4 * i], lam[4 * i + 1], lam[4 * i + 2], lam[4 * i + 3]);
        float r = rgb[3 * i], g = rgb[3 * i + 1], b = rgb[3 * i + 2];
        S4 s = mode == 0 ? eval_bounded(coef_bounded(T, r, g, b), l) : (mode == 1 ? eval_scaled(coef_unbounded(T, r, g, b), l) : eval_illuminant(coef_illuminant(T, r, g, b), l));
        out[4 * i] = s.x;
        out[4 * i + 1] = s.y;
        out[4 * i + 2] = s.z;
        out[4 * i + 3] = s.w;
    }
}
__global__ void k_test_light_bvh(DScene sc, int n, const float* p3, const float* n3, const float* u, int* out_light, float* out_pmf, const int* query, float* out_qpmf) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        v3 p = mk3(p3[3 * i], p3[3 * i + 1], p3[3 * i + 2]), nn = mk3(n3[3 * i], n3[3 * i + 1], n3[3 * i + 2]);
        float pmf;
        unsigned vis = 0;
        out_light[i] = bvh_sample_light(sc, p, nn, u[i], pmf, vis);
        out_pmf[i] = pmf;
        if (query) out_qpmf[i] = bvh_pmf(sc, p, nn, query[i], vis);
    }
}

// ---------------------------------------------------------------------------------------------------
// launch wrappers (called from hk_api.cpp)
// ---------------------------------------------------------------------------------------------------
namespace hk {

static inline int grid_for(int n, int block, int cap) {
    long g = ((long)n + block - 1) / block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (int)g;
}

// blocks per CU that are actually resident for a kernel (occupancy API, capped): the wave-segment loop makes any
// grid size correct, so the grid is sized to residency instead of oversubscribing and paying a tail round.
template <class K>
static int resident_blocks(K kernel, int block, int n_cu, int cap_per_cu) {
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, block, 0) != hipSuccess || per_cu < 1) per_cu = 1;
    if (per_cu > cap_per_cu) per_cu = cap_per_cu;
    return per_cu * n_cu;
}
static int clamp_blocks(int blocks, const DPathState& st) {
    int maxb = st.n_waves / 4;
    return blocks > maxb ? maxb : blocks;
}

void launch_camera(hipStream_t s, int n_cu, const DPathState& st, const DFrame& fr, const DTables& T, const DFilter& f, const DCamera& c, const DSobol& sob, int initial_medium) {
    static int blocks = resident_blocks(k_camera, 256, n_cu, 8);
    hipLaunchKernelGGL(k_camera, dim3(clamp_blocks(blocks, st)), dim3(256), 0, s, st, fr, T, f, c, sob, initial_medium);
}
void launch_trace(hipStream_t s, int n_cu, const DPathState& st, const DScene& sc, const DTables& T, const DFrame& fr, int depth, DStats* stats) {
    static int b0 = resident_blocks(k_trace<false>, HK_TRACE_BLOCK, n_cu, 8), b1 = resident_blocks(k_trace<true>, HK_TRACE_BLOCK, n_cu, 8);
    if (fr.count_nodes)
        hipLaunchKernelGGL(k_trace<true>, dim3(clamp_blocks(b1, st)), dim3(HK_TRACE_BLOCK), 0, s, st, sc, T, fr, depth, stats);
    else
        hipLaunchKernelGGL(k_trace<false>, dim3(clamp_blocks(b0, st)), dim3(HK_TRACE_BLOCK), 0, s, st, sc, T, fr, depth, stats);
}
void launch_shadow(hipStream_t s, int n_cu, const DPathState& st, const DScene& sc, const DTables& T, const DFrame& fr, int depth, DStats* stats) {
    const bool lean = sc.all_opaque && sc.n_media == 0;
#define HK_SHADOW_LAUNCH(C, L)                                                                                                  \
    {                                                                                                                           \
        static int blocks = resident_blocks(k_shadow<C, L>, HK_TRACE_BLOCK, n_cu, 8);                                            \
        hipLaunchKernelGGL((k_shadow<C, L>), dim3(clamp_blocks(blocks, st)), dim3(HK_TRACE_BLOCK), 0, s, st, sc, T, depth, stats); \
    }
    if (fr.count_nodes) {
        if (lean) HK_SHADOW_LAUNCH(true, true) else HK_SHADOW_LAUNCH(true, false)
    } else {
        if (lean) HK_SHADOW_LAUNCH(false, true) else HK_SHADOW_LAUNCH(false, false)
    }
#undef HK_SHADOW_LAUNCH
}
void launch_medium(hipStream_t s, int n_cu, const DPathState& st, const DScene& sc, const DTables& T, const DFrame& fr, const DSobol& sob, int depth, DStats* stats) {
    static int blocks = resident_blocks(k_medium<false>, 256, n_cu, 8);
    hipLaunchKernelGGL(k_medium<false>, dim3(clamp_blocks(blocks, st)), dim3(256), 0, s, st, sc, T, fr, sob, depth, stats);
}
void launch_detect_camera_medium(hipStream_t s, const DPathState& st, const DScene& sc, float x, float y, float z, DStats* stats) {
    hipLaunchKernelGGL(k_detect_camera_medium, dim3(1), dim3(64), 0, s, st, sc, x, y, z, stats);
}
void launch_escaped(hipStream_t s, int n_cu, const DPathState& st, const DScene& sc, const DTables& T, int depth) {
    static int blocks = resident_blocks(k_escaped, 256, n_cu, 8);
    hipLaunchKernelGGL(k_escaped, dim3(clamp_blocks(blocks, st)), dim3(256), 0, s, st, sc, T, depth);
}
void launch_shade(hipStream_t s, int n_cu, int kind, const DPathState& st, const DScene& sc, const DTables& T, const DFrame& fr, const DSobol& sob, int depth, int first_kind, DStats* stats) {
#define HK_SHADE_CASE(K)                                                                                                          \
    case K: {                                                                                                                     \
        static int blocks = resident_blocks(k_shade<K>, 256, n_cu, 8);                                                            \
        hipLaunchKernelGGL(k_shade<K>, dim3(clamp_blocks(blocks, st)), dim3(256), 0, s, st, sc, T, fr, sob, depth, first_kind, stats); \
    } break;
    switch (kind) {
        HK_SHADE_CASE(HK_MAT_MATTE)
        HK_SHADE_CASE(HK_MAT_MIRROR)
        HK_SHADE_CASE(HK_MAT_GLASS)
        HK_SHADE_CASE(HK_MAT_CONDUCTOR)
        HK_SHADE_CASE(HK_MAT_COATED_DIFFUSE)
        HK_SHADE_CASE(HK_MAT_THIN_DIELECTRIC)
        HK_SHADE_CASE(HK_MAT_DIFFUSE_TRANSMISSION)
        HK_SHADE_CASE(HK_MAT_COATED_DIFFUSE_TRANSMISSION)
        HK_SHADE_CASE(HK_MAT_COATED_CONDUCTOR)
        default: {
            static int blocks = resident_blocks(k_shade<HK_MAT_FALLBACK>, 256, n_cu, 8);
            hipLaunchKernelGGL(k_shade<HK_MAT_FALLBACK>, dim3(clamp_blocks(blocks, st)), dim3(256), 0, s, st, sc, T, fr, sob, depth, first_kind, stats);
        } break;
    }
#undef HK_SHADE_CASE
}
void launch_film(hipStream_t s, const DPathState& st, const DFrame& fr, const DTables& T, void* accum, bool f64) {
    int g = grid_for(fr.n_pixels_padded, 256, 4096);
    if (f64)
        hipLaunchKernelGGL(k_film<double>, dim3(g), dim3(256), 0, s, st, fr, T, (double*)accum);
    else
        hipLaunchKernelGGL(k_film<float>, dim3(g), dim3(256), 0, s, st, fr, T, (float*)accum);
}
void launch_finalize(hipStream_t s, const void* accum, bool f64, float* out, int w, int h) {
    int g = grid_for(w * h, 256, 4096);
    if (f64)
        hipLaunchKernelGGL(k_finalize<double>, dim3(g), dim3(256), 0, s, (const double*)accum, out, w, h);
    else
        hipLaunchKernelGGL(k_finalize<float>, dim3(g), dim3(256), 0, s, (const float*)accum, out, w, h);
}
template <int KIND>
__device__ void test_bsdf_one(const DScene& sc, const DTables& T, const DMaterial& m, int mode, bool regularize, v3 wo, v3 wi, v3 ns, S4 lambda, v2 u, float uc, float* r) {
    if (mode == 0) {
        BSDFSample b = sample_bsdf<KIND>(sc, T, m, wo, ns, mk2(0.0f, 0.0f), lambda, u, uc, regularize);
        r[0] = b.wi.x, r[1] = b.wi.y, r[2] = b.wi.z;
        r[3] = b.f.x, r[4] = b.f.y, r[5] = b.f.z, r[6] = b.f.w;
        r[7] = b.pdf, r[8] = b.is_specular ? 1.0f : 0.0f, r[9] = b.eta_scale;
    } else {
        float pdf;
        S4 f = eval_bsdf<KIND>(sc, T, m, wo, wi, ns, mk2(0.0f, 0.0f), lambda, pdf);
        r[0] = f.x, r[1] = f.y, r[2] = f.z, r[3] = f.w, r[4] = pdf;
        r[5] = r[6] = r[7] = r[8] = r[9] = 0.0f;
    }
}
__global__ void k_test_bsdf(DScene sc, DTables T, int mode, int mat_idx, int regularize, int n, const float* wo, const float* wi, const float* ns, const float* lambda,
                            const float* u, const float* uc, float* out) {
    const DMaterial& m = sc.materials[mat_idx];
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    v3 o = mk3(wo[3 * i], wo[3 * i + 1], wo[3 * i + 2]), d = mk3(wi[3 * i], wi[3 * i + 1], wi[3 * i + 2]), nn = mk3(ns[3 * i], ns[3 * i + 1], ns[3 * i + 2]);
    S4 l = s4(lambda[4 * i], lambda[4 * i + 1], lambda[4 * i + 2], lambda[4 * i + 3]);
    v2 uu = mk2(u[2 * i], u[2 * i + 1]);
    float* r = out + 10 * (size_t)i;
#define HK_TB_CASE(K) \
    case K: test_bsdf_one<K>(sc, T, m, mode, regularize != 0, o, d, nn, l, uu, uc[i], r); break;
    switch (m.kind) {
        HK_TB_CASE(HK_MAT_MATTE)
        HK_TB_CASE(HK_MAT_MIRROR)
        HK_TB_CASE(HK_MAT_GLASS)
        HK_TB_CASE(HK_MAT_CONDUCTOR)
        HK_TB_CASE(HK_MAT_COATED_DIFFUSE)
        HK_TB_CASE(HK_MAT_THIN_DIELECTRIC)
        HK_TB_CASE(HK_MAT_DIFFUSE_TRANSMISSION)
        HK_TB_CASE(HK_MAT_COATED_DIFFUSE_TRANSMISSION)
        HK_TB_CASE(HK_MAT_COATED_CONDUCTOR)
        default: test_bsdf_one<HK_MAT_FALLBACK>(sc, T, m, mode, regularize != 0, o, d, nn, l, uu, uc[i], r); break;
    }
    }
#undef HK_TB_CASE
}
__global__ void k_test_light(DScene sc, DTables T, int mode, int light_idx, int n, const float* p3, const float* in3, const float* lambda, float* out) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        S4 l = s4(lambda[4 * i], lambda[4 * i + 1], lambda[4 * i + 2], lambda[4 * i + 3]);
        v3 a = mk3(in3[3 * i], in3[3 * i + 1], in3[3 * i + 2]);
        float* r = out + 12 * (size_t)i;
        for (int k = 0; k < 12; ++k) r[k] = 0.0f;
        if (mode == 0) {
            LightSample ls = sample_light(sc, T, sc.lights[light_idx - 1], mk3(p3[3 * i], p3[3 * i + 1], p3[3 * i + 2]), l, mk2(a.x, a.y));
            r[0] = ls.wi.x, r[1] = ls.wi.y, r[2] = ls.wi.z, r[3] = ls.pdf;
            r[4] = ls.Li.x, r[5] = ls.Li.y, r[6] = ls.Li.z, r[7] = ls.Li.w;
            r[8] = ls.p_light.x, r[9] = ls.p_light.y, r[10] = ls.p_light.z, r[11] = ls.is_delta ? 1.0f : 0.0f;
        } else {
            S4 Le = s4(0.0f);
            float pdf = 0.0f;
            for (int li = 0; li < sc.n_lights; ++li) {
                const DLight& L = sc.lights[li];
                if (L.kind == HK_LIGHT_AMBIENT) Le = Le + L.scale * light_spectrum(L, l);
                if (L.kind == HK_LIGHT_ENVIRONMENT) {
                    float4 t = env_eval(sc.envmaps[L.Le_tex], a);
                    Le = Le + eval_illuminant(coef_illuminant(T, t.x * L.Le_rgba[0], t.y * L.Le_rgba[1], t.z * L.Le_rgba[2]), l);
                    pdf = pdf + env_pdf_li(sc.envmaps[L.Le_tex], a);
                }
            }
            r[0] = Le.x, r[1] = Le.y, r[2] = Le.z, r[3] = Le.w, r[4] = pdf;
        }
    }
}
void launch_test_trace(hipStream_t s, const DScene& sc, int n, const float* o, const float* d, const float* tmax, float* t, int* prim, float* uv) {
    hipLaunchKernelGGL(k_test_trace, dim3(grid_for(n, HK_TRACE_BLOCK, 1280)), dim3(HK_TRACE_BLOCK), 0, s, sc, n, o, d, tmax, t, prim, uv);
}
void launch_test_sobol(hipStream_t s, const DTables& T, const DSobol& sob, int n, const int* px, const int* py, const int* si, const int* dim, float* o1, float* o2) {
    hipLaunchKernelGGL(k_test_sobol, dim3(grid_for(n, 256, 1024)), dim3(256), 0, s, T, sob, n, px, py, si, dim, o1, o2);
}
void launch_test_camera(hipStream_t s, const DTables& T, const DFilter& f, const DCamera& c, const DSobol& sob, int height, int n, const int* px, const int* py, const int* si,
                        float* out) {
    hipLaunchKernelGGL(k_test_camera, dim3(grid_for(n, 256, 1024)), dim3(256), 0, s, T, f, c, sob, height, n, px, py, si, out);
}
void launch_test_uplift(hipStream_t s, const DTables& T, int mode, int n, const float* rgb, const float* lam, float* out) {
    hipLaunchKernelGGL(k_test_uplift, dim3(grid_for(n, 256, 1024)), dim3(256), 0, s, T, mode, n, rgb, lam, out);
}
void launch_test_light_bvh(hipStream_t s, const DScene& sc, int n, const float* p, const float* nn, const float* u, int* ol, float* op, const int* q, float* oq) {
    hipLaunchKernelGGL(k_test_light_bvh, dim3(grid_for(n, 256, 1024)), dim3(256), 0, s, sc, n, p, nn, u, ol, op, q, oq);
}
void launch_test_bsdf(hipStream_t s, const DScene& sc, const DTables& T, int mode, int mat_idx, int regularize, int n, const float* wo, const float* wi, const float* ns,
                      const float* lambda, const float* u, const float* uc, float* out) {
    hipLaunchKernelGGL(k_test_bsdf, dim3(grid_for(n, 64, 4096)), dim3(64), 0, s, sc, T, mode, mat_idx, regularize, n, wo, wi, ns, lambda, u, uc, out);
}
void launch_test_light(hipStream_t s, const DScene& sc, const DTables& T, int mode, int light_idx, int n, const float* p3, const float* in3, const float* lambda, float* out) {
    hipLaunchKernelGGL(k_test_light, dim3(grid_for(n, 64, 4096)), dim3(64), 0, s, sc, T, mode, light_idx, n, p3, in3, lambda, out);
}

}  // namespace hk

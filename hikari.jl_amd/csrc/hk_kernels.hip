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
#include <type_traits>
#include <cstdlib>
#include <cstdio>

#include "hikari_mi355x.h"
#include "hk_device.h"

using namespace hkd;

namespace {

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }

// Wave-private segmented queues: wave segment w owns entries [w*wave_cap, (w+1)*wave_cap) of every queue and the count word
// counts[..][w].  A path never leaves the segment that generated its camera ray, so compaction is pure ballot/popcount
// arithmetic: no atomics, no cursors, deterministic order.  (A single global queue counter serialises at ~88 atomics/us on this
// chip: 40 k wave-pushes per launch cost ~0.45 ms per kernel.)
struct WaveQ {
    uint32_t* base;
    int count;  // wave-uniform
};
// Wave index of the launch.  Everything that ties data to an XCD (static segment strides, ticket counters) assumes 4-wave blocks dealt
// round-robin over the 8 XCDs: wave g runs on XCD (g / 4) % 8.  A 16-wave block (k_trace_lean with its node cache) numbers its
// waves so that this still holds: it stands for the four 4-wave blocks vb = ((b / 8) * 4 + w / 4) * 8 + b % 8 (a bijection when
// the grid is a multiple of 8 blocks, which one block per CU is).
__device__ __forceinline__ int global_wave() {
    const int w = (int)(threadIdx.x >> 6), b = (int)blockIdx.x;
    if (blockDim.x == 1024 && (gridDim.x & 7) == 0) return ((((b >> 3) * 4 + (w >> 2)) * 8 + (b & 7)) << 2) + (w & 3);
    if (blockDim.x == 512 && (gridDim.x & 7) == 0) return ((((b >> 3) * 2 + (w >> 2)) * 8 + (b & 7)) << 2) + (w & 3);   // an 8-wave block (k_light_select_pool): two such 4-wave blocks
    return b * (int)(blockDim.x >> 6) + w;
}
__device__ __forceinline__ int physical_waves() { return (int)(gridDim.x * (blockDim.x >> 6)); }
// A kernel is launched with as many physical waves as are resident for ITS register/LDS budget; each physical wave walks the
// virtual wave segments.  Surface scenes walk them with a static stride (the grid is clamped to a divisor of W, see
// clamp_blocks: residency differs per kernel — 12 .. 28 waves per CU — and a stride that does not divide W leaves a tail round
// with a fraction of the waves busy).  Scenes with media, whose work per segment is wildly uneven (a tile that looks into a
// cloud vs one that sees the sky), hand the segments out through tickets: +9 % on the cloud config, -16 % on the sky config.
// One ticket = one atomic, and ONE WORD takes ~88 atomics/us on this chip: with a single ticket word a launch over W = 24 k
// segments cost 0.3 ms even when every segment was empty — the launch floor of the deep bounces of the cloud config in round 1
// (k_escaped / k_scatter / k_shade min 316 - 328 us).  Handing out several segments per ticket is no way out (4 per ticket made the
// shadow walk 48 % slower: tail).  The ticket is therefore split HK_TICKET_WAYS = 64 ways: counter k, in its own 256-byte line
// (so another memory channel), hands out the segments k, k + K, k + 2K, ...; a wave starts at counter (wave mod K) and moves on
// to the next live counter when one is exhausted, never to come back.  Same one-segment granularity and the same stealing
// behaviour, 1/K of the serialisation.
// WORK LISTS.  At the deep bounces most segments of a queue are empty (an open scene loses half of its paths per bounce), and a
// launch that visits all W segments pays a count-word round trip (plus a ticket) per empty one.  After every producer the host
// launches k_segment_lists, which writes, per queue, the ascending list of its non-empty segments and their number; consumers
// iterate over the list — statically (entry w, w + waves, ...) or by ticket — and never see an empty segment.
#ifndef HK_TICKET_REFRESH
#define HK_TICKET_REFRESH 0   // see seg_next
#endif
#ifndef HK_TICKET_OPEN_SHARED
#define HK_TICKET_OPEN_SHARED 1
#endif
struct SegTickets {
    int* cnt;                   // HK_TICKET_WAYS (= 64: one per lane) counters, HK_TICKET_STRIDE ints apart
    unsigned long long alive;   // counters not yet seen exhausted (wave-uniform)
    int k0;                     // where this wave starts looking
    const int* list;            // non-empty segments of the consumed queue, ascending (null: every segment 0 .. n - 1)
    int n;                      // entries to hand out
    int pos, step;              // static mode: next entry of this wave, stride (step == 0: ticket mode)
    unsigned long long own;     // ticket mode: the counters whose segments the static stride would give to this wave's XCD
    bool share;                 // ticket mode: exhausted counters are published in one shared word (seg_next)
};
// Static stride: block b runs on XCD b % 8 and its wave w takes the list entries 4b + w, 4b + w + waves, ...: the records one kernel
// writes are read by the next kernel from the SAME XCD's L2 while they are still there (the per-XCD L2s do not share).  Tickets keep
// that affinity: counter k hands out entries k, k + 64, ..., all congruent to k mod 32, so counter k "belongs" to XCD (k / 4) % 8; a wave
// draws from the eight counters of its own XCD first and steals from the others only when those are exhausted (sky scene: the
// deep, sparse bounces fit in L2 — shade +12 % slower with XCD-blind tickets).
__device__ __forceinline__ unsigned long long xcd_counters() {
    int x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    x &= 7;
    return (0xfull << (4 * x)) | (0xfull << (4 * x + 32));
}
__device__ __forceinline__ int seg_per_way(int n_segments, int k) { return (n_segments - k + HK_TICKET_WAYS - 1) / HK_TICKET_WAYS; }
__device__ __forceinline__ const int* seg_list_ptr(const DPathState& st, int depth, int q) { return st.seg_list + (size_t)(depth * Q_COUNT + q) * st.n_waves; }
// dynamic: all 64 counters are inspected with ONE parallel load (lane k reads counter k): an exhausted launch costs a wave one
// memory round trip and no atomic at all.  depth < 0: no list, every segment (the camera kernel).
__device__ __forceinline__ SegTickets seg_open(const DPathState& st, int* cnt, bool dynamic, int depth, int q) {
    SegTickets it;
    it.cnt = cnt;
    const bool lists = depth >= 0 && st.small_pass == 0;
    it.list = lists ? seg_list_ptr(st, depth, q) : nullptr;
    it.n = lists ? st.seg_list_n[depth * Q_COUNT + q] : st.n_waves;
    if (st.small_pass) dynamic = false;
    it.k0 = global_wave() & (HK_TICKET_WAYS - 1);
    it.pos = global_wave();
    it.step = dynamic ? 0 : physical_waves();
    it.alive = 0ull;
    it.own = 0ull;
    it.share = st.ticket_share != 0;
    if (dynamic) {
        it.own = xcd_counters();
        const int lane = lane_id();
#if HK_TICKET_OPEN_SHARED
        if (it.share) {   // the shared word says which counters are used up: one 8-byte read instead of 64 lines per wave and launch
            const unsigned long long m = __hip_atomic_load(reinterpret_cast<unsigned long long*>(cnt + 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)m), hi = __builtin_amdgcn_readfirstlane((unsigned)(m >> 32));
            it.alive = __ballot(0 < seg_per_way(it.n, lane)) & ~(((unsigned long long)hi << 32) | lo);
        } else
#endif
        {
            const int v = __hip_atomic_load(cnt + lane * HK_TICKET_STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            it.alive = __ballot(v < seg_per_way(it.n, lane));
        }
    }
    return it;
}
__device__ __forceinline__ int seg_entry(const SegTickets& it, int i) { return it.list ? it.list[i] : i; }
__device__ __forceinline__ int seg_next(SegTickets& it, int n_segments) {   // -> segment index, or n_segments when none is left
    if (it.step != 0) {
        const int i = it.pos;
        if (i >= it.n) return n_segments;
        it.pos = i + it.step;
        return seg_entry(it, i);
    }
    while (it.alive != 0ull) {
        // first live counter at or after k0 (cyclically)
        const unsigned long long cand = (it.alive & it.own) != 0ull ? (it.alive & it.own) : it.alive;
        const unsigned long long rot = it.k0 == 0 ? cand : ((cand >> it.k0) | (cand << (64 - it.k0)));
        const int k = (it.k0 + __ffsll((long long)rot) - 1) & (HK_TICKET_WAYS - 1);
        const int per_k = seg_per_way(it.n, k);
        int t = 0;
        if (lane_id() == 0) t = atomicAdd(it.cnt + k * HK_TICKET_STRIDE, 1);
        t = __builtin_amdgcn_readfirstlane(t);
        it.k0 = k;
        if (t < per_k) return seg_entry(it, k + t * HK_TICKET_WAYS);
        // counters only grow: exhausted once, exhausted for good.  At the end of a launch every wave finds that out one failed atomic at
        // a time (up to 63 serial round trips).  HK_TICKET_REFRESH=1 re-reads all 64 counters with one parallel load after a failure
        // instead: the deep bounces of a 32-spp Cornell frame lose 20 - 30 us per launch, the 256-spp frame and the many-light frame gain
        // 0.5 %, but the 64 extra line reads per wave get in the way of the remaining atomics where launches are short and many — the
        // cloud frame +3.6 %, Cornell at 64 spp +3 % — so it stays off (round 4, interleaved on one box).  What does pay where launches are
        // short and many is ONE shared word of "exhausted" bits per ticket (DPathState::ticket_share): cloud frame -1.8 % (644 -> 633 ms,
        // three interleaved pairs); with seg_open reading that word instead of the 64 counters another -0.4 %, Cornell -0.5 %,
        // many-light -0.3 %.
#if HK_TICKET_REFRESH
        const int v = __hip_atomic_load(it.cnt + lane_id() * HK_TICKET_STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        it.alive &= __ballot(v < seg_per_way(it.n, lane_id()));
#endif
        if (it.share) {   // one shared word per ticket (the unused second 128-B half of counter 0's 256 bytes): a wave that finds counter k
                          // exhausted says so and learns, with the same atomic, what the others have found
            unsigned long long m = 0ull;
            if (lane_id() == 0) m = __hip_atomic_fetch_or(reinterpret_cast<unsigned long long*>(it.cnt + 32), 1ull << k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)m), hi = __builtin_amdgcn_readfirstlane((unsigned)(m >> 32));
            it.alive &= ~(((unsigned long long)hi << 32) | lo);
        }
        it.alive &= ~(1ull << k);
    }
    return n_segments;
}
// depth / q: the queue the kernel consumes (its work list); depth < 0: all segments
#define HK_FOR_EACH_WAVE_SEGMENT(gw, st, ticket, depth, q)                                                        \
    for (SegTickets gw##_t = seg_open(st, ticket, (st).dynamic_segments != 0, depth, q); gw##_t.cnt; gw##_t.cnt = nullptr) \
        for (int gw = seg_next(gw##_t, (st).n_waves); gw < (st).n_waves; gw = seg_next(gw##_t, (st).n_waves))
// the same loop over a prepared source (seg_open's result, or seg_single: the stage bodies are shared with k_small_pass)
#define HK_FOR_EACH_SEGMENT_FROM(gw, st, tickets)                                                                  \
    for (SegTickets gw##_t = (tickets); gw##_t.cnt; gw##_t.cnt = nullptr)                                         \
        for (int gw = seg_next(gw##_t, (st).n_waves); gw < (st).n_waves; gw = seg_next(gw##_t, (st).n_waves))
// exactly the segment g (static mode: the counter pointer only says "open")
__device__ __forceinline__ SegTickets seg_single(const DPathState& st, int g) {
    SegTickets it;
    it.cnt = st.tickets;
    it.alive = 0ull;
    it.k0 = 0;
    it.list = nullptr;
    it.n = g + 1;
    it.pos = g;
    it.step = 1 << 30;
    it.own = 0ull;
    it.share = false;
    return it;
}
// A kernel whose lanes leave nothing behind in their segment (the shadow kernels: results go to L[slot]) does not have to drain its
// lanes at the end of every segment: it asks for one segment after another and the per-lane refill simply continues with the next
// segment's entries.  The only drain left is the one at the end of the launch (the shadow walk of the cloud config ran a third of
// its lane-slots empty in the tails of its ~1000-entry segments).
typedef SegTickets SegStream;
__device__ __forceinline__ SegStream stream_open(const DPathState& st, int* ticket, bool force_dynamic, int depth, int q) {
    return seg_open(st, ticket, force_dynamic || st.dynamic_segments != 0, depth, q);
}
__device__ __forceinline__ int stream_next(SegStream& s, int n_segments) { return seg_next(s, n_segments); }
// ticket words: row = bounce depth (row max_depth + 1: camera / film), column = kernel
enum { TK_TRACE = 0, TK_TRACK = 1, TK_SHADOW = 2, TK_ESCAPED = 3, TK_SCATTER = 4, TK_SHADE0 = 5, TK_SELECT = 1, TK_CAMERA = 0, TK_FILM = 1 };   // TK_SELECT shares TK_TRACK's column: k_light_select only runs in scenes without media
__device__ __forceinline__ int* ticket_ptr(const DPathState& st, int row, int col) { return st.tickets + (size_t)(row * HK_TICKET_COLS + col) * (HK_TICKET_WAYS * HK_TICKET_STRIDE); }
__device__ __forceinline__ WaveQ wq_open(uint32_t* q, const DPathState& st, int gw) { return WaveQ{q + (size_t)gw * st.wave_cap, 0}; }
__device__ __forceinline__ void wq_push(WaveQ& q, uint32_t value, bool active) {
    unsigned long long mask = __ballot(active);
    if (active) q.base[q.count + __popcll(mask & ((1ull << lane_id()) - 1ull))] = value;
    q.count += __popcll(mask);
}
// beta / r_u / r_l of generation entry p; `ones`: the depth-0 records of a scene without media hold the constant 1 implicitly
__device__ __forceinline__ S4 ld_throughput(const float4* arr, size_t p, bool ones) { return ones ? s4(1.0f) : ld4(&arr[p]); }
// r_u / r_l of generation entry p and the MIS weights of a shadow record, in either representation (DPathState::compact)
__device__ __forceinline__ S4 ld_ru(const DPathGen& g, size_t p, bool ones, bool compact) { return (ones || compact) ? s4(1.0f) : ld4(&g.r_u[p]); }
// (compact: r_l is one float and lives in the fourth word of the record's ray_d — the reference's ray.time, which this path never reads)
__device__ __forceinline__ S4 ld_rl(const DPathGen& g, size_t p, bool ones, bool compact) {
    if (ones) return s4(1.0f);
    if (compact) return s4(reinterpret_cast<const float*>(g.ray_d)[4 * p + 3]);
    return ld4(&g.r_l[p]);
}
// r_u / r_l of a record whose ray_d has ALREADY been stored with r_l.x in its fourth word when `compact` (the writers do that themselves)
__device__ __forceinline__ void st_ru_rl(const DPathGen& g, size_t p, S4 r_u, S4 r_l, bool compact) {
    if (!compact) {
        st4(&g.r_u[p], r_u);
        st4(&g.r_l[p], r_l);
    }
}
__device__ __forceinline__ void mul_rl(const DPathGen& g, size_t p, bool ones, float f, bool compact) {   // r_l[p] *= f
    if (compact) {
        float* w = reinterpret_cast<float*>(g.ray_d) + 4 * p + 3;
        *w = (ones ? 1.0f : *w) * f;
    } else
        st4(&g.r_l[p], ld_throughput(g.r_l, p, ones) * f);
}
__device__ __forceinline__ void st_shadow_weights(const DPathState& st, size_t rec, S4 ru, S4 rl) {
    if (st.compact)
        reinterpret_cast<float2*>(st.sh_ru)[rec] = make_float2(ru.x, rl.x);
    else {
        st4(&st.sh_ru[rec], ru);
        st4(&st.sh_rl[rec], rl);
    }
}
// COMPACT is a compile-time fact in the shadow kernels: k_shadow and k_shadow_walk<.., 0> only run in scenes without media
template <bool COMPACT>
__device__ __forceinline__ void ld_shadow_weights(const DPathState& st, size_t rec, S4& ru, S4& rl) {
    if (COMPACT) {
        const float2 w = reinterpret_cast<const float2*>(st.sh_ru)[rec];
        ru = s4(w.x);
        rl = s4(w.y);
    } else {
        ru = ld4(&st.sh_ru[rec]);
        rl = ld4(&st.sh_rl[rec]);
    }
}
// Streaming stores / loads of path and shadow records (written once, read once by a later kernel): marked non-temporal so that they
// do not push the register-spill lines of the resident waves out of the L2s (HK_NT=0 at build time: plain accesses)
#ifndef HK_NT
#define HK_NT 1
#endif
typedef uint32_t hk_u2v __attribute__((ext_vector_type(2)));
HKD void stream_st(float4* p, float4 v) {
#if HK_NT
    __builtin_nontemporal_store((hk_f4v){v.x, v.y, v.z, v.w}, (hk_f4v*)p);
#else
    *p = v;
#endif
}
HKD void stream_st(float4* p, S4 v) { stream_st(p, make_float4(v.x, v.y, v.z, v.w)); }
HKD void stream_st(uint2* p, uint2 v) {
#if HK_NT
    __builtin_nontemporal_store((hk_u2v){v.x, v.y}, (hk_u2v*)p);
#else
    *p = v;
#endif
}
HKD void stream_st(float* p, float v) {
#if HK_NT
    __builtin_nontemporal_store(v, p);
#else
    *p = v;
#endif
}
HKD void stream_st(uint32_t* p, uint32_t v) {
#if HK_NT
    __builtin_nontemporal_store(v, p);
#else
    *p = v;
#endif
}
HKD float4 stream_ld(const float4* p) {
#if HK_NT
    const hk_f4v v = __builtin_nontemporal_load((const hk_f4v*)p);
    return make_float4(v.x, v.y, v.z, v.w);
#else
    return *p;
#endif
}
// The four wavelengths of a path never change after the camera drew them (TerminateSecondary is not part of the reference's path), so
// they are kept ONCE, by camera slot (lambda_s, which the film kernel needs anyway), instead of travelling with every generation
// record: 16 B less to write per continuing vertex and per camera ray, the same 16 B to read (slots of a queue ascend, so the reads
// stay nearly dense).  HK_LAMBDA_BY_SLOT=0 (hk_types.h) builds the round-2 layout (a copy in every generation) for A/B runs.
// Measured (round 3): films bit-identical; k_camera -5 %, k_shade unchanged (it is not bound by its bytes alone: DESIGN.md §5), 5 GB less
// path state at 164 M paths in flight.
__device__ __forceinline__ S4 ld_lambda(const DPathState& st, const DPathGen& g, size_t p, uint32_t pslot) {
#if HK_LAMBDA_BY_SLOT
    const float4 v = stream_ld(&st.lambda_s[pslot]);
#else
    const float4 v = stream_ld(&g.lambda[p]);
#endif
    return s4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void st_lambda(const DPathGen& g, size_t p, S4 lambda) {
#if !HK_LAMBDA_BY_SLOT
    stream_st(&g.lambda[p], lambda);
#endif
}
// lean generation records (DPathState::meta32 / const_origin): the record's meta word(s) and origin through one pair of accessors
__device__ __forceinline__ uint2 ld_meta(const DPathState& st, const DPathGen& g, size_t p, int depth) {
    if (st.meta32) {
        const uint32_t w = reinterpret_cast<const uint32_t*>(g.meta)[p];
        return make_uint2((uint32_t)depth | (((w >> 30) & 1u) << 8) | ((w >> 31) << 9), w & 0x3fffffffu);
    }
    return g.meta[p];
}
__device__ __forceinline__ void st_meta(const DPathState& st, const DPathGen& g, size_t p, uint32_t flags, uint32_t slot) {
    if (st.meta32)
        stream_st(reinterpret_cast<uint32_t*>(g.meta) + p, slot | (((flags >> 8) & 1u) << 30) | (((flags >> 9) & 1u) << 31));
    else
        stream_st(&g.meta[p], make_uint2(flags, slot));
}
// const_origin: generation 0 keeps ONE origin per segment, at the segment's first entry `seg` (k_camera's lane with position 0 writes it),
// and every depth-0 reader of that segment loads that entry — one hot line instead of 16 B per path; no branch, no arithmetic repeated
__device__ __forceinline__ float4 ld_ray_o(const DPathState& st, const DPathGen& g, size_t p, int depth, size_t seg) {
    return stream_ld(&g.ray_o[(depth == 0 && st.const_origin) ? seg : p]);
}
// dense append of a whole record: the position the pushing lanes get inside the segment (count + rank among the pushing lanes)
struct WavePos {
    int count;  // wave-uniform: entries already in the segment
};
__device__ __forceinline__ int wp_push(WavePos& q, bool active) {   // returns this lane's position (valid where active)
    unsigned long long mask = __ballot(active);
    int pos = q.count + __popcll(mask & ((1ull << lane_id()) - 1ull));
    q.count += __popcll(mask);
    return pos;
}
__device__ __forceinline__ int* count_ptr(const DPathState& st, int depth, int q, int gw) { return st.counters + ((size_t)(depth * Q_COUNT + q) * st.n_waves + gw); }
__device__ __forceinline__ void wq_close(const WaveQ& q, int* cnt) {
    if (lane_id() == 0) *cnt = q.count;
}

#ifdef HK_DEBUG_UTIL
#define HK_DBG_DECL unsigned long long dbg_[32] = {0};
#define HK_DBG(i, active) do { dbg_[2 * (i)] += 64; dbg_[2 * (i) + 1] += __popcll(__ballot(active)); } while (0)
#define HK_DBG_FLUSH(stats) do { if (lane_id() == 0) for (int k_ = 0; k_ < 32; ++k_) (stats)->dbg[k_] += dbg_[k_]; } while (0)
#else
#define HK_DBG_DECL
#define HK_DBG(i, active) do { } while (0)
#define HK_DBG_FLUSH(stats) do { } while (0)
#endif
// per-wave statistics rows (summed on the host): plain read-modify-write, the row belongs to this wave
__device__ __forceinline__ void wave_add(unsigned long long* dst, unsigned v) {
    unsigned s = v;
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off);
    if (lane_id() == 0 && s) *dst += (unsigned long long)s;
}

// Path slot of a pass -> (pixel slot, pass-sample k).  The samples of a pixel are ADJACENT slots (slot = pixel slot * S + k): a wave of
// camera rays is 64 samples of one pixel, the per-(pixel, dimension) sampler tables are read by neighbouring lanes, and k_film walks
// a contiguous run.  Division by the pass's sample count through a host-made reciprocal (DFrame::s_mul / s_shr, exact below 2^30).
__device__ __forceinline__ void split_slot(const DFrame& fr, uint32_t slot, int& pix, int& k) {
    const uint32_t q = (uint32_t)(((uint64_t)slot * fr.s_mul) >> fr.s_shr);
    pix = (int)q;
    k = (int)(slot - q * (uint32_t)fr.samples_in_pass);
}
__device__ __forceinline__ void slot_to_pixel(const DFrame& fr, int slot_in_sample, int& px, int& py, bool& inside) {
    int tile = slot_in_sample >> 6, l = slot_in_sample & 63;
    int tx = tile % fr.tiles_x, ty = tile / fr.tiles_x;
    px = fr.x0 + tx * 8 + (l & 7);
    py = fr.y0 + ty * 8 + (l >> 3);
    inside = px < fr.x1 && py < fr.y1;
}

}  // namespace

// Pixel-digit table of the ZSobol index (DSobol::hi_table): entry (row, pixel slot) = the permuted base-4 digits above the
// sample bits, for the dimension of that row.  Built once per film size / sampler seed.
__global__ void __launch_bounds__(256) k_sobol_table(DSobol sob, DFrame fr, uint2* table, int rows) {
    const long total = (long)rows * fr.n_pixels_padded;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        int row = (int)(i / fr.n_pixels_padded), pix = (int)(i - (long)row * fr.n_pixels_padded);
        int px, py;
        bool inside;
        slot_to_pixel(fr, pix, px, py, inside);
        const int dim = sobol_row_dim(row);
        uint64_t m = (left_shift2((uint64_t)(uint32_t)(py + 1)) << 1) | left_shift2((uint64_t)(uint32_t)(px + 1));
        uint64_t morton = m << sob.log2_spp;
        const int pow2 = sob.log2_spp & 1;
        const uint64_t dmix = 0x55555555ull * (uint64_t)(int64_t)dim;
        uint64_t hi = zsobol_digits(morton, dmix, pow2, sob.n_base4_digits - 1, zsobol_first_pixel_digit(sob.log2_spp));
        table[i] = make_uint2((uint32_t)(hi >> sob.log2_spp), zsobol_top_perms(morton, dmix, sob.log2_spp));
    }
}

// Work lists (see SegTickets): the non-empty segments of a queue and their number.
struct SegQueues {
    int n;
    int depth[HK_MAX_KINDS + 6], q[HK_MAX_KINDS + 6];
};
// grid (queues, HK_LIST_SPLIT): block y writes the non-empty segments of its contiguous share, ascending, behind those of the shares
// before it — whose number it counts itself (a few dozen loads per thread), so the list is globally ascending (dense queues give
// the identity, which keeps the static stride's segment -> wave -> XCD assignment) and no block waits for another.
#define HK_LIST_SPLIT 8
__global__ void __launch_bounds__(1024) k_segment_lists(DPathState st, SegQueues qs) {
    __shared__ int wave_total[16];
    const int d = qs.depth[blockIdx.x], q = qs.q[blockIdx.x];
    const int* __restrict__ cnt = st.counters + (size_t)(d * Q_COUNT + q) * st.n_waves;
    int* __restrict__ out = st.seg_list + (size_t)(d * Q_COUNT + q) * st.n_waves;
    const int lane = (int)(threadIdx.x & 63), wave = (int)(threadIdx.x >> 6);
    const int share = (st.n_waves + HK_LIST_SPLIT - 1) / HK_LIST_SPLIT;
    // both clamped: for small segment counts ceil(n / 8) * 7 can exceed n (n = 12: share 2, block 7 would start at 14 and count the NEXT
    // queue's words into its prefix)
    const int begin = (int)blockIdx.y * share < st.n_waves ? (int)blockIdx.y * share : st.n_waves, end = begin + share < st.n_waves ? begin + share : st.n_waves;
    // the non-empty segments before this block's share
    int mine = 0;
    for (int i = (int)threadIdx.x; i < begin; i += 1024) mine += cnt[i] != 0 ? 1 : 0;
    for (int off = 32; off > 0; off >>= 1) mine += __shfl_down(mine, off);
    if (lane == 0) wave_total[wave] = mine;
    __syncthreads();
    int base = 0;
    for (int w = 0; w < 16; ++w) base += wave_total[w];
    for (int start = begin; start < end; start += 1024) {
        const int i = start + (int)threadIdx.x;
        const bool nz = i < end && cnt[i] != 0;
        const unsigned long long m = __ballot(nz);
        __syncthreads();
        if (lane == 0) wave_total[wave] = __popcll(m);
        __syncthreads();
        int before = 0, total = 0;
        for (int w = 0; w < 16; ++w) {
            const int t = wave_total[w];
            before += w < wave ? t : 0;
            total += t;
        }
        if (nz) out[base + before + __popcll(m & ((1ull << lane) - 1ull))] = i;
        base += total;
    }
    if (blockIdx.y == HK_LIST_SPLIT - 1 && threadIdx.x == 0) st.seg_list_n[d * Q_COUNT + q] = base;
}

// Sample-bit table (DSobol::lo_table): the low log2_spp bits of the permuted index for sample indices base + j * stride, j < count.
// One thread per 16 entries.  With stride 1 and base a multiple of 16 the sixteen share every digit above the last two, the
// permutation of the second-to-last digit is chosen by the digits above it (one hash for all sixteen) and that of the last digit by
// the digits above IT (one hash per four): 9 hashes per 16 entries instead of 4 per entry (Cornell 800^2, 49 rows, 256 spp: 16 GB).
__global__ void __launch_bounds__(256) k_sobol_lo_table(DSobol sob, DFrame fr, uint16_t* table, int rows, int base, int stride, int count) {
    const long groups = count >> 4;
    const long total = (long)rows * fr.n_pixels_padded * groups;
    const uint32_t mask = (1u << sob.log2_spp) - 1u;
    const int pow2 = sob.log2_spp & 1;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long rp = i / groups;
        const int q = (int)(i - rp * groups);
        const int row = (int)(rp / fr.n_pixels_padded), pix = (int)(rp - (long)row * fr.n_pixels_padded);
        int px, py;
        bool inside;
        slot_to_pixel(fr, pix, px, py, inside);
        const int dim = sobol_row_dim(row);
        const uint64_t dmix = 0x55555555ull * (uint64_t)(int64_t)dim;
        const uint64_t m = ((left_shift2((uint64_t)(uint32_t)(py + 1)) << 1) | left_shift2((uint64_t)(uint32_t)(px + 1))) << sob.log2_spp;
        const uint2 e = sob.hi_table[(size_t)row * sob.hi_stride + pix];
        uint32_t v[16];
        const int s0 = base + 16 * q * stride;
        if (stride == 1 && pow2 == 0 && (s0 & 15) == 0 && sob.log2_spp >= 4) {
            const uint64_t morton = m | (uint64_t)(uint32_t)s0;
            const uint32_t upper = (uint32_t)zsobol_sample_index_cached(morton, dim, sob.log2_spp, e.x, e.y) & mask & ~15u;
            const int p1 = zsobol_perm_index(morton >> 4, dmix);
            for (int d1 = 0; d1 < 4; ++d1) {
                const uint32_t hi = upper | ((uint32_t)zsobol_permute_digit(p1, d1) << 2);
                const int p0 = zsobol_perm_index((morton | ((uint64_t)d1 << 2)) >> 2, dmix);
                for (int d0 = 0; d0 < 4; ++d0) v[4 * d1 + d0] = hi | (uint32_t)zsobol_permute_digit(p0, d0);
            }
        } else {
            for (int t = 0; t < 16; ++t) v[t] = (uint32_t)zsobol_sample_index_cached(m | (uint64_t)(uint32_t)(s0 + t * stride), dim, sob.log2_spp, e.x, e.y) & mask;
        }
        uint4* out = reinterpret_cast<uint4*>(table) + 2 * i;
        out[0] = make_uint4(v[0] | (v[1] << 16), v[2] | (v[3] << 16), v[4] | (v[5] << 16), v[6] | (v[7] << 16));
        out[1] = make_uint4(v[8] | (v[9] << 16), v[10] | (v[11] << 16), v[12] | (v[13] << 16), v[14] | (v[15] << 16));
    }
}

// ---------------------------------------------------------------------------------------------------
// K1: camera rays (volpath.jl:125-205).  One thread per path slot of the pass.
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ void camera_body(const DPathState& st, const DFrame& fr, const DTables& T, const DFilter& flt, const DCamera& cam, const DSobol& sob, const SegTickets& src) {
    const int total = fr.n_pixels_padded * fr.samples_in_pass;
    const int n_chunks = total >> 6;
    const DPathGen g0 = st.gen[0];
    HK_FOR_EACH_SEGMENT_FROM(gw, st, src) {
    WavePos out{0};
    const size_t seg = (size_t)gw * st.wave_cap;
    // wave w generates chunks w, w+W, w+2W, ... (64 consecutive slots = samples of one pixel, or of 64 / S pixels; interleaved across waves for load balance)
    for (int chunk = gw; chunk < n_chunks; chunk += st.n_waves) {
        int slot = chunk * 64 + lane_id();
        int pix, k;
        split_slot(fr, (uint32_t)slot, pix, k);
        int px, py;
        bool active;
        slot_to_pixel(fr, pix, px, py, active);
        const size_t p = seg + (size_t)wp_push(out, active);   // generation 0 holds the pass's paths in camera order, film padding squeezed out
        if (active) {
            int sample_idx = fr.first_sample + k * fr.sample_stride;
            int x = px + 1, y = py + 1;  // 1-based pixel coordinates (Q1)
            SobolCtx sc = sobol_ctx(sob, T.sobol, x, y, sample_idx, pix, k);
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
            if (!st.const_origin || p == seg) stream_st(&g0.ray_o[p], make_float4(ro.x, ro.y, ro.z, INF_F));   // (a pinhole camera's rays all start at one point: ld_ray_o)
            // beta = r_u = r_l = 1 at depth 0 (volpath.jl:190-197): in scenes without media nothing changes them before the first
            // shading event, so they are not stored and the depth-0 readers substitute the constant (ld_throughput); the compact
            // records of a grey medium keep r_l = 1 beside the direction
            stream_st(&g0.ray_d[p], make_float4(rd.x, rd.y, rd.z, (st.compact && !fr.implicit_ones) ? 1.0f : 0.0f));
            st_lambda(g0, p, lambda);
            if (!fr.implicit_ones) {
                st4(&g0.beta[p], s4(1.0f));
                st_ru_rl(g0, p, s4(1.0f), s4(1.0f), st.compact != 0);
            }
            st_meta(st, g0, p, (uint32_t)(*st.initial_medium + 1) << 16, (uint32_t)slot);
            stream_st(&st.lambda_s[slot], lambda);
            stream_st(&st.L[slot], s4(0.0f));
            stream_st(&st.filter_w[slot], fw);
        }
    }
    if (lane_id() == 0) *count_ptr(st, 0, Q_RAY, gw) = out.count;
    }
}
__global__ void __launch_bounds__(256) k_camera(DPathState st, DFrame fr, DTables T, DFilter flt, DCamera cam, DSobol sob, int initial_medium) {
    camera_body(st, fr, T, flt, cam, sob, seg_open(st, ticket_ptr(st, st.ticket_rows - 1, TK_CAMERA), st.dynamic_segments != 0, -1, 0));
}

// Copies the first min(NC, sc.n_nodes) nodes into the block's LDS (see NodeCache) and waits for the whole block.
template <int NC, int BLOCK, bool QN = false>
HKD NodeCache node_cache_fill(const DScene& sc, float4* __restrict__ box, int2* __restrict__ child) {
    NodeCache c;
    c.box = (const lds_float4*)box, c.child = (const lds_int2*)child;
    c.tri = nullptr, c.nt = 0;
    c.nc = sc.n_nodes < NC ? sc.n_nodes : NC;
    if (NC > 0) {
        for (int i = threadIdx.x; i < c.nc; i += BLOCK) {
            if (QN) {   // quantised tree: the LDS copy holds the grid coordinates as floats (node_step's arithmetic is the same for both arms)
                const uint4* qp = reinterpret_cast<const uint4*>(sc.qnodes) + 2 * (size_t)i;
                const uint4 P = qp[0], Q = qp[1];
                box[i] = make_float4((float)(P.x & 0xffffu), (float)(P.x >> 16), (float)(P.y & 0xffffu), (float)(P.y >> 16));
                box[NC + i] = make_float4((float)(P.z & 0xffffu), (float)(P.z >> 16), (float)(P.w & 0xffffu), (float)(P.w >> 16));
                box[2 * NC + i] = make_float4((float)(Q.x & 0xffffu), (float)(Q.x >> 16), (float)(Q.y & 0xffffu), (float)(Q.y >> 16));
                child[i] = make_int2((int)Q.z, (int)Q.w);
                continue;
            }
            const float4* np = reinterpret_cast<const float4*>(sc.nodes) + 4 * (size_t)i;
            const float4 D = np[3];
            box[i] = np[0], box[NC + i] = np[1], box[2 * NC + i] = np[2];
            child[i] = make_int2(__float_as_int(D.x), __float_as_int(D.y));
        }
        __syncthreads();
    }
    return c;
}

// The scenes of the media kernels are often a handful of boxes (the cloud config: 24 triangles, 7 nodes): k_trace keeps the first NC
// nodes AND the first NT leaf triangles in LDS, so such a cast touches no global memory at all (cloud trace 87 -> 65 ms).  Not the
// shadow walk: the same cache inside k_shadow_walk cost it 28 % (its registers are full: 168 at 3 waves per SIMD).
#define HK_MEDIA_NC 64
#define HK_MEDIA_NT 80    // 32 KB of stacks + 7.25 KB of cache: 4 blocks per CU, as without it
template <int NC, int NT, int BLOCK>
HKD NodeCache scene_cache_fill(const DScene& sc, float4* __restrict__ box, int2* __restrict__ child, float4* __restrict__ tri) {
    NodeCache c = node_cache_fill<NC, BLOCK>(sc, box, child);
    c.tri = (const lds_float4*)tri;
    c.nt = sc.n_tris < NT ? sc.n_tris : NT;
    for (int i = threadIdx.x; i < c.nt; i += BLOCK) {
        const float4* tp = sc.leaf_tris + 3 * (size_t)i;
        tri[i] = tp[0], tri[NT + i] = tp[1], tri[2 * NT + i] = tp[2];
    }
    __syncthreads();
    return c;
}

// ---------------------------------------------------------------------------------------------------
// K3: closest-hit traversal + classification (intersection.jl:188-269).
// ---------------------------------------------------------------------------------------------------
template <bool COUNT>
__global__ void __launch_bounds__(HK_TRACE_BLOCK) k_trace(DPathState st, DScene sc, DTables T, DFrame fr, int depth, DStats* stats) {
    __shared__ int lds_stack[(HK_TRACE_BLOCK / 64) * HK_LDS_STACK * 64];
    __shared__ float4 lds_box[3 * HK_MEDIA_NC];
    __shared__ int2 lds_child[HK_MEDIA_NC];
    __shared__ float4 lds_tri[3 * HK_MEDIA_NT];
    int* stack = lds_stack + (threadIdx.x >> 6) * (HK_LDS_STACK * 64);
    const NodeCache cache = scene_cache_fill<HK_MEDIA_NC, HK_MEDIA_NT, HK_TRACE_BLOCK>(sc, lds_box, lds_child, lds_tri);
    const int lane = lane_id();
    unsigned n_nodes = 0, n_tris = 0, n_casts = 0, n_hits = 0;
    HK_FOR_EACH_WAVE_SEGMENT(gw, st, ticket_ptr(st, depth, TK_TRACE), depth, Q_RAY) {
    const DPathGen g = st.gen[depth & 1];
    const uint32_t seg = (uint32_t)gw * (uint32_t)st.wave_cap;   // the segment's live rays are entries seg .. seg + n - 1 of this generation
    const int n = *count_ptr(st, depth, Q_RAY, gw);
    WaveQ q_escaped = wq_open(st.escaped_q, st, gw);
    WaveQ q_medium = wq_open(st.medium_q, st, gw);
    int kind_count[HK_MAX_KINDS];
#pragma unroll
    for (int k = 0; k < HK_MAX_KINDS; ++k) kind_count[k] = 0;
    for (int base = 0; base < n; base += 64) {
        int i = base + lane;
        bool active = i < n;
        uint32_t slot = seg + (uint32_t)(active ? i : 0);   // generation index of the path
        int kind = -1;  // -1 none, -2 escaped, >= 0 material kind
        bool in_medium = false;
        if (active && sc.n_media > 0 && (g.meta[slot].x >> 16) != 0u) {
            // ray travels inside a medium: one cast (no alpha test, intersection.jl:198-221), then delta tracking
            // (k_medium) decides whether the stored surface hit is ever reached
            in_medium = true;
            float4 O = g.ray_o[slot], D = g.ray_d[slot];
            bool dummy;
            ++n_casts;
            HitRec h = traverse<0, COUNT, HK_MEDIA_NC, HK_MEDIA_NT>(sc, mk3(O.x, O.y, O.z), mk3(D.x, D.y, D.z), O.w, stack, lane, n_nodes, n_tris, dummy, cache);
            if (h.prim >= 0) {
                ++n_hits;
                st.hit[slot] = make_float4(h.t, __int_as_float(h.prim), h.u, h.v);
                const DTriMeta hm = sc.meta[h.prim];
                st.mat_id[slot] = sc.mis[hm.mi].material | (hm.arealight > 0 ? HK_MAT_EMISSIVE_BIT : 0);
            } else
                st.hit[slot] = make_float4(INF_F, __int_as_float(-1), 0.0f, 0.0f);
        }
        wq_push(q_medium, slot, in_medium);
        if (active && !in_medium) {
            float4 O = g.ray_o[slot], D = g.ray_d[slot];
            v3 ro = mk3(O.x, O.y, O.z), rd = mk3(D.x, D.y, D.z);
            float tmax = O.w;
            // alpha-test loop: alpha-killed surfaces are skipped without consuming depth (<= 16 casts)
            for (int it = 0; it < 16; ++it) {
                bool dummy;
                ++n_casts;
                HitRec h = traverse<0, COUNT, HK_MEDIA_NC, HK_MEDIA_NT>(sc, ro, rd, it == 0 ? tmax : INF_F, stack, lane, n_nodes, n_tris, dummy, cache);
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
                st.mat_id[slot] = mat | (meta.arealight > 0 ? HK_MAT_EMISSIVE_BIT : 0);
                if (it > 0) g.ray_o[slot] = make_float4(ro.x, ro.y, ro.z, INF_F);  // origin after alpha skips
                break;
            }
        }
        // ballot-compact into this wave's per-kind queue segments
        wq_push(q_escaped, slot, kind == -2);
        unsigned long long pending = __ballot(kind >= 0);
        while (pending) {
            int src = __ffsll((long long)pending) - 1;
            const int k = __builtin_amdgcn_readlane(kind, src);   // src is wave-uniform: the kind and every count below stay in scalar registers
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
// K3 / K10 for scenes without media whose surfaces are all opaque (the common case): while-while traversal with wave-private
// dynamic fetch.  The step count per ray varies by 3-4x inside a wave of incoherent rays; instead of letting finished lanes
// idle until the slowest ray of the batch is done, a lane whose ray is finished keeps its result until the next flush, where
// results are classified / pushed with ballots and idle lanes pull the next rays of the wave's own queue segment.
// Same arithmetic and same per-ray result as traverse<>() (closest hit: min (t, prim); shadow: any hit).
// ---------------------------------------------------------------------------------------------------
// refill once this many lanes are idle: 12 was the optimum with pixel-tile waves; since a wave holds the samples of a few neighbouring
// pixels (rays that finish together more often) 24 - 32 is (many-light trace -4 %, shadow -3 %; flat from 24 to 40)
#ifndef HK_TRACE_MIN_IDLE
#define HK_TRACE_MIN_IDLE 24
#endif
// the closest-hit KERNEL of scenes whose tree sits in the 16-wave block's LDS cache (Cornell, sky: BVHs <= 16 deep) refills at 32 idle lanes —
// round 6, A B | B A on one box: k_trace_lean 34.45 -> 33.6 ms per Cornell frame (40: 34.15, 48: 35.4), sky 19.95 -> 19.65; the deep-tree
// instantiation of the 10^6-triangle scene keeps 24 (32: +1.2 %, 40: +2.6 %), and so do the any-hit kernel (32: +-0) and the small pass
#ifndef HK_CLOSEST_MIN_IDLE_LDS
#define HK_CLOSEST_MIN_IDLE_LDS 32
#endif
// leaf phases wait until this many lanes hold a leaf (0: never wait), as long as the kernel can still refill its idle lanes
#ifndef HK_ANYHIT_LEAF_MIN
#define HK_ANYHIT_LEAF_MIN 16
#endif
#ifndef HK_CLOSEST_LEAF_MIN
#define HK_CLOSEST_LEAF_MIN 0
#endif
// closest hit with the whole tree in LDS: the node loop of a round ends once A * (lanes still descending) < lanes waiting with a leaf
#ifndef HK_LEAF_BREAK_A
#define HK_LEAF_BREAK_A 2
#endif
// POSTPONED LEAVES (round 5; Aila & Laine's "speculative traversal").  A lane that reaches a leaf while its stack still holds nodes does
// not wait for the wave's next leaf phase: it keeps the leaf in `pend` and goes on descending from the popped node; only a lane that
// reaches a SECOND leaf (or has nothing left to descend) waits.  Node steps and leaf phases both run fuller; the price is one leaf's
// worth of stale t_best (a few more nodes visited).  The hit does not depend on the order in which leaves are tested (closest hit:
// minimum of (t, prim); any hit: whether one exists), so films are bit-identical (test_lean_traversal_parity, ab_bitwise.py).
// MEASURED AND SWITCHED OFF (interleaved on one box, films bit-identical): the stale t_best costs more node visits than the fuller phases
// return — Cornell trace +1.5 %, shadow +5 %; 10^6 triangles trace +7 %, shadow +4 %; sky trace +10 %.
#ifndef HK_POSTPONE_CLOSEST
#define HK_POSTPONE_CLOSEST 0
#endif
#ifndef HK_POSTPONE_ANYHIT
#define HK_POSTPONE_ANYHIT 0
#endif
#ifndef HK_NODE_UNROLL
#define HK_NODE_UNROLL 3   // node steps per evaluation of the node loop's exit rule (ballots, popcounts, the branch): 1 / 2 / 3 steps — Cornell trace 32.8 / 32.0 / 31.6 ms,
#endif                     // 10^6 triangles 0.531 / 0.518 / 0.517 s (interleaved on one box, films bit-identical); lanes that reach a leaf sit out the rest of the group
struct LaneRay {   // per-lane traversal state
    v3 o, d;
    RaySlab rs;
    float t_max;
    HitRec best;
    int cur, sp;
    int pend;   // a postponed leaf reference, or DONE (0x80000000) for none; cur == DONE implies pend == DONE
    bool any;   // lane_ray_round<.., MIXED>: this lane's ray is an any-hit (shadow) ray among closest-hit rays (trace_shadow_body)
};
template <bool QN = false>
HKD void lane_ray_start(LaneRay& r, const DScene& sc, v3 o, v3 d, float t_max) {
    r.o = o;
    r.d = d;
    r.t_max = t_max;
    r.best.t = t_max;
    r.best.prim = -1;
    r.best.u = r.best.v = 0.0f;
    r.rs = QN ? ray_slab_grid(sc, o, d) : ray_slab(o, d);
    r.sp = 0;
    r.cur = sc.n_tris == 0 ? (int)0x80000000 : sc.root_ref;
    r.pend = (int)0x80000000;
    r.any = false;
}
// one while-while round for the lanes with `active`: inner nodes until every such lane holds a leaf (or is done), then the leaves.
// ANYHIT: the first accepted triangle ends the ray (cur = DONE, best.prim >= 0).
// MIXED (with ANYHIT = false): closest-hit and any-hit rays share the wave, LaneRay::any says which a lane holds; the round's rules are the
// closest-hit kernel's, a lane's leaf test is its own kind's.
template <bool ANYHIT, bool COUNT, int NC = 0, bool POSTPONE = (ANYHIT ? HK_POSTPONE_ANYHIT != 0 : HK_POSTPONE_CLOSEST != 0), bool MIXED = false, bool QN = false>
HKD void lane_ray_round(LaneRay& r, bool active, const DScene& sc, int* __restrict__ stack, int lane, unsigned& n_nodes, unsigned& n_tris, const NodeCache& cache = NodeCache(),
                        bool may_wait = false
#ifdef HK_DEBUG_UTIL
                        , unsigned long long* dbg_ = nullptr
#endif
                        ) {
    const int DONE = (int)0x80000000;
    // ballots of ONE comparison each, combined as scalar masks: the ballot of a compound condition costs two VALU instructions
    // (v_cndmask + v_cmp) on top of the comparisons, and this loop is bound by instruction issue
    const unsigned long long act_m = __builtin_amdgcn_ballot_w64(active);
    // The rule that ends the node loop weighs what a leaf costs against a node step.  With the WHOLE tree in LDS (closest hit in the
    // Cornell box) a node step is cheap and the leaf phase — 46 % of the lanes under the 1 : 1 rule, up to four triangles each — is
    // where the lane-slots go: the nodes run on until they are outnumbered 2 : 1 (trace -5 %, frame +1.7 %).  Trees that reach into
    // global memory keep 1 : 1 (sky scene at 2 : 1: trace +1 %; 10^6 triangles: +3 %), and so does the any-hit kernel (+-0 either way).
    const int break_a = (!ANYHIT && NC > 0 && sc.n_nodes <= NC) ? HK_LEAF_BREAK_A : 1;
    for (;;) {
        const unsigned long long in_nodes = act_m & __builtin_amdgcn_ballot_w64(r.cur >= 0);
        if (in_nodes == 0ull) break;
        // stragglers: once fewer lanes are still descending than are waiting with a leaf, test the leaves first — the descending
        // lanes keep their state and go on in the next round.  (Running the node loop until the LAST lane holds a leaf cost
        // half of the traversal time in the 10^6-triangle scene: trace 0.94 -> 0.46 s, shadow 0.30 -> 0.19 s.)
        const int n_desc = __builtin_popcountll(in_nodes), n_leaf = __builtin_popcountll(act_m & __builtin_amdgcn_ballot_w64((unsigned)r.cur > 0x80000000u));   // cur < 0 and not DONE
        if (break_a * n_desc < n_leaf) break;
        // (ending the round here as soon as the caller's refill threshold is reached — rays that finish inside the node loop leave idle
        // lanes behind — fills the node steps better, 67 -> 72 % in the any-hit kernel, and still loses: the flush / refill it triggers
        // more often costs more than the lanes it feeds.  Cornell trace +7 %, shadow +9 %, 10^6 triangles +3 %.)
#ifdef HK_DEBUG_UTIL
        if (dbg_) HK_DBG(0, active && r.cur >= 0);          // node steps: lanes that descend
#endif
#pragma unroll
        for (int rep = 1; rep < HK_NODE_UNROLL; ++rep)
            if (active && r.cur >= 0) {
                if (COUNT) ++n_nodes;
                node_step<NC, QN>(sc, r.rs, r.best.t, stack, lane, r.cur, r.sp, cache);
            }
        if (active && r.cur >= 0) {
            if (COUNT) ++n_nodes;
            node_step<NC, QN>(sc, r.rs, r.best.t, stack, lane, r.cur, r.sp, cache);
            if (POSTPONE) {
                if (r.cur == DONE) {   // nothing left to descend: the postponed leaf (if any) is what is left of this ray
                    r.cur = r.pend;
                    r.pend = DONE;
                } else if (r.cur < 0 && r.pend == DONE && r.sp > 0) {   // a first leaf, and more to descend: postpone it
                    r.pend = r.cur;
                    --r.sp;
                    r.cur = stack[r.sp * 64 + lane];
                }
            }
        }
    }
    // The leaf phase is the expensive half of a round (up to four triangles), and it used to run for however few lanes held a leaf: in the
    // any-hit kernel, where most rays finish in the node loop without ever reaching one, 14 % of its lane-slots did work (HK_DEBUG_UTIL).
    // While the caller can still refill (`may_wait`), the few leaf holders now keep their leaf and wait: the caller's refill brings new
    // rays, and the leaves are tested once LEAF_MIN lanes hold one (Cornell: leaf phases 14 % -> 32 % full and 57 % fewer of them, k_shadow -5 %;
    // 8 or 20 instead of 16: no better).  (After the node loop either nothing descends or fewer lanes descend
    // than hold leaves: with fewer than LEAF_MIN <= 20 leaf holders at least 24 lanes are idle, so the refill is certain to happen.)
    constexpr int LEAF_MIN = ANYHIT ? (NC > 0 ? HK_ANYHIT_LEAF_MIN : 0) : HK_CLOSEST_LEAF_MIN;   // the deep-tree instantiations (NC == 0: 10^6 triangles) lose 1 % by waiting
    static_assert(LEAF_MIN == 0 || 64 - 2 * LEAF_MIN >= HK_TRACE_MIN_IDLE, "waiting leaf holders must leave enough idle lanes for the caller's refill to trigger");
    // (with postponed leaves the wait only looks at the lanes that are STUCK with a leaf in `cur`: the descending ones take their
    // postponed leaf along)
    if (LEAF_MIN > 0 && may_wait && __builtin_popcountll(act_m & __builtin_amdgcn_ballot_w64((unsigned)r.cur > 0x80000000u)) < LEAF_MIN) return;
    const bool has_pend = POSTPONE && r.pend != DONE;
#ifdef HK_DEBUG_UTIL
    if (dbg_) HK_DBG(1, active && (has_pend || (r.cur < 0 && r.cur != DONE)));   // leaf phase: lanes that hold a leaf
#endif
#ifdef HK_DEBUG_UTIL
    if (dbg_) {   // triangle iterations of the leaf phase: the wave runs max(count) of them, sum(count) lane-slots work (probe 6: closest hit, 7: any hit — an upper bound there)
        const bool holds = active && (has_pend || (r.cur < 0 && r.cur != DONE));
        int c = holds ? ((~(has_pend ? r.pend : r.cur)) & 7) + 1 : 0, mx = c, sm = c;
        for (int off = 32; off > 0; off >>= 1) {
            mx = max(mx, __shfl_xor(mx, off));
            sm += __shfl_xor(sm, off);
        }
        dbg_[ANYHIT ? 8 : 12] += 64ull * (unsigned)mx;
        dbg_[ANYHIT ? 9 : 13] += (unsigned)sm;
    }
#endif
    if (active && (has_pend || (r.cur < 0 && r.cur != DONE))) {
        int ref = ~(has_pend ? r.pend : r.cur);
        int first = ref >> 3, count = (ref & 7) + 1;
        bool stop = false;
        if (!ANYHIT && !MIXED) {
            // closest hit tests every triangle of the leaf, so the next one is fetched while this one is tested (trace -4 % in the
            // Cornell box, -2 % elsewhere); the any-hit loop below usually stops early and is better off without (shadow +1 %)
            const float4* tp = sc.leaf_tris + 3 * (size_t)first;
            float4 T0 = tp[0], T1 = tp[1], T2 = tp[2];
            for (int i = 0; i < count; ++i) {
                float4 N0 = T0, N1 = T1, N2 = T2;
                if (i + 1 < count) N0 = tp[3 * i + 3], N1 = tp[3 * i + 4], N2 = tp[3 * i + 5];
                asm volatile("" ::"v"(T0.x), "v"(T0.y), "v"(T0.z), "v"(T0.w));
                if (COUNT) ++n_tris;
                float t, u, v;
                if (intersect_triangle(r.o, r.d, r.t_max, mk3(T0.x, T0.y, T0.z), mk3(T1.x, T1.y, T1.z), mk3(T2.x, T2.y, T2.z), t, u, v)) {
                    int prim = __float_as_int(T0.w);
                    if (r.best.prim < 0 || t < r.best.t || (t == r.best.t && prim < r.best.prim)) {
                        r.best.t = t;
                        r.best.prim = prim;
                        r.best.u = u;
                        r.best.v = v;
                    }
                }
                T0 = N0, T1 = N1, T2 = N2;
            }
        } else
        for (int i = 0; i < count && !stop; ++i) {
            const float4* tp = sc.leaf_tris + 3 * (size_t)(first + i);
            float4 T0 = tp[0], T1 = tp[1], T2 = tp[2];
            // all three loads are issued here: left alone, the compiler sinks the v0 load behind the det != 0 test and
            // every triangle pays a second memory round trip
            asm volatile("" ::"v"(T0.x), "v"(T0.y), "v"(T0.z), "v"(T0.w));
            if (COUNT) ++n_tris;
            float t, u, v;
            if (intersect_triangle(r.o, r.d, r.t_max, mk3(T0.x, T0.y, T0.z), mk3(T1.x, T1.y, T1.z), mk3(T2.x, T2.y, T2.z), t, u, v)) {
                int prim = __float_as_int(T0.w);
                if (MIXED ? r.any : ANYHIT) {
                    r.best.t = t;
                    r.best.prim = prim;
                    stop = true;
                } else if (r.best.prim < 0 || t < r.best.t || (t == r.best.t && prim < r.best.prim)) {
                    r.best.t = t;
                    r.best.prim = prim;
                    r.best.u = u;
                    r.best.v = v;
                }
            }
        }
        if (stop) {
            r.cur = DONE;
            if (POSTPONE) r.pend = DONE;
        } else if (has_pend)
            r.pend = DONE;   // (cur — a node to descend, or a second leaf for the next phase — stays)
        else if (r.sp > 0) {
            --r.sp;
            r.cur = stack[r.sp * 64 + lane];
        } else
            r.cur = DONE;
    }
}

enum { LR_EMPTY = 0, LR_ACTIVE = 1 };

// STACK: LDS stack entries per lane.  The stack holds at most one entry per inner level, so a BVH of depth <= 16 (every scene
// but the 10^6-triangle one) runs with half the LDS: 16 KB per block instead of 32, and LDS stops limiting residency.
template <bool COUNT, int NC, bool QN = false, int MIN_IDLE = HK_TRACE_MIN_IDLE>
__device__ __forceinline__ void trace_lean_body(const DPathState& st, const DScene& sc, int depth, DStats* stats, const SegTickets& src, int* __restrict__ stack, const NodeCache& cache) {
    const int lane = lane_id();
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    const int DONE = (int)0x80000000;
    unsigned n_nodes = 0, n_tris = 0, n_casts = 0, n_hits = 0;
    HK_DBG_DECL
    HK_FOR_EACH_SEGMENT_FROM(gw, st, src) {
    const DPathGen g = st.gen[depth & 1];
    const uint32_t seg = (uint32_t)gw * (uint32_t)st.wave_cap;   // the rays of this segment: entries seg .. seg + n - 1, read in order (no index queue)
    const int n = *count_ptr(st, depth, Q_RAY, gw);
    WaveQ q_escaped = wq_open(st.escaped_q, st, gw);
    int kind_count[HK_MAX_KINDS];
#pragma unroll
    for (int k = 0; k < HK_MAX_KINDS; ++k) kind_count[k] = 0;
    int cursor = 0;
    int state = LR_EMPTY;
    uint32_t slot = 0;
    LaneRay r;
    r.cur = r.pend = DONE;
    for (;;) {
        const unsigned long long run_m = __ballot(state == LR_ACTIVE && r.cur != DONE);
        if (run_m == 0ull || (64 - __popcll(run_m) >= MIN_IDLE && cursor < n)) {
            // ---- flush: classify the finished rays, push them to the escaped / material-kind queues ----
            HK_DBG(8, state == LR_ACTIVE && r.cur == DONE);   // flushes: lanes that deliver a finished ray
            int kind = -1;
            if (state == LR_ACTIVE && r.cur == DONE) {
                if (r.best.prim < 0)
                    kind = -2;
                else {
                    ++n_hits;
                    const DTriMeta hm = sc.meta[r.best.prim];
                    int mat = sc.mis[hm.mi].material;
                    if (sc.materials[mat].kind == HK_MAT_MIX) {   // MixMaterial is resolved here so the queue is sorted by the final kind
                        float w = 1.0f - r.best.u - r.best.v;
                        mat = resolve_mix_material(sc, mat, r.o + r.d * r.best.t, -r.d, uv_at(sc, r.best.prim, w, r.best.u, r.best.v));
                    }
                    kind = sc.materials[mat].kind;
                    if (kind == HK_MAT_MIX) kind = HK_MAT_FALLBACK;
                    stream_st(&st.hit[slot], make_float4(r.best.t, __int_as_float(r.best.prim), r.best.u, r.best.v));
                    stream_st(reinterpret_cast<uint32_t*>(st.mat_id) + slot, (uint32_t)(mat | (hm.arealight > 0 ? HK_MAT_EMISSIVE_BIT : 0)));
                }
                state = LR_EMPTY;
            }
            wq_push(q_escaped, slot, kind == -2);
            unsigned long long pending = __ballot(kind >= 0);
            while (pending) {
                int src = __ffsll((long long)pending) - 1;
                const int k = __builtin_amdgcn_readlane(kind, src);   // src is wave-uniform: the kind and every count below stay in scalar registers
                bool mine = kind == k;
                unsigned long long m = __ballot(mine);
                int cnt = 0;
#pragma unroll
                for (int kk = 0; kk < HK_MAX_KINDS; ++kk) cnt = (kk == k) ? kind_count[kk] : cnt;
                if (mine) stream_st(&st.mat_q[((size_t)k * st.n_waves + gw) * st.wave_cap + cnt + __popcll(m & lt_mask)], slot);
                int add = __popcll(m);
#pragma unroll
                for (int kk = 0; kk < HK_MAX_KINDS; ++kk) kind_count[kk] += (kk == k) ? add : 0;
                pending &= ~m;
            }
            // ---- refill ----
            const unsigned long long want = __ballot(state == LR_EMPTY);
            const int avail = n - cursor;
            const int rank = __popcll(want & lt_mask);
            if (state == LR_EMPTY && rank < avail) {
                slot = seg + (uint32_t)(cursor + rank);
                float4 O = ld_ray_o(st, g, slot, depth, seg), D = stream_ld(&g.ray_d[slot]);
                ++n_casts;
                lane_ray_start<QN>(r, sc, mk3(O.x, O.y, O.z), mk3(D.x, D.y, D.z), O.w);
                state = LR_ACTIVE;
            }
            const int want_n = __popcll(want);
            cursor += want_n < avail ? want_n : (avail > 0 ? avail : 0);
            if (__ballot(state == LR_ACTIVE) == 0ull) break;
        }
#ifdef HK_DEBUG_UTIL
        HK_DBG(2, state == LR_ACTIVE && r.cur != DONE);      // rounds: lanes with a ray in flight
        lane_ray_round<false, COUNT, NC, HK_POSTPONE_CLOSEST != 0, false, QN>(r, state == LR_ACTIVE && r.cur != DONE, sc, stack, lane, n_nodes, n_tris, cache, cursor < n, dbg_);
#else
        lane_ray_round<false, COUNT, NC, HK_POSTPONE_CLOSEST != 0, false, QN>(r, state == LR_ACTIVE && r.cur != DONE, sc, stack, lane, n_nodes, n_tris, cache, cursor < n);
#endif
    }
    wq_close(q_escaped, count_ptr(st, depth, Q_ESCAPED, gw));
    if (lane == 0) {
        *count_ptr(st, depth, Q_MEDIUM, gw) = 0;
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
    HK_DBG_FLUSH(stats);
}
template <bool COUNT, int STACK, int BLOCK = HK_TRACE_BLOCK, int NC = 0, bool QN = false>
__global__ void __launch_bounds__(BLOCK) k_trace_lean(DPathState st, DScene sc, int depth, DStats* stats) {
    __shared__ int lds_stack[(BLOCK / 64) * STACK * 64];
    __shared__ float4 lds_box[NC > 0 ? 3 * NC : 1];
    __shared__ int2 lds_child[NC > 0 ? NC : 1];
    int* stack = lds_stack + (threadIdx.x >> 6) * (STACK * 64);
    const NodeCache cache = node_cache_fill<NC, BLOCK, QN>(sc, lds_box, lds_child);
    trace_lean_body<COUNT, NC, QN, (NC >= 1024 ? HK_CLOSEST_MIN_IDLE_LDS : HK_TRACE_MIN_IDLE)>(st, sc, depth, stats, seg_open(st, ticket_ptr(st, depth, TK_TRACE), st.dynamic_segments != 0, depth, Q_RAY), stack, cache);
}

// ---------------------------------------------------------------------------------------------------
// K4: delta tracking with null collisions (delta-tracking.jl:79-453), after k_trace, for the paths that travel inside a
// medium.  The collision count per path is wildly uneven (0 .. 1000s), so the wave does not walk its queue 64 entries at
// a time: every lane is a little state machine that takes ONE step per iteration (next majorant segment, or one tentative
// collision) and, when its path is finished, waits for the next refill, where finished paths are routed with
// ballot/popcount pushes (scatter queue -> k_scatter, escaped queue, material-kind queues with Mix resolved) and idle lanes
// pull the next entries of the wave's own queue segment.  Still no atomics: queue, cursor and counts are wave-private.
// Arithmetic and RNG consumption per path are exactly those of the sequential loop.
// ---------------------------------------------------------------------------------------------------
#ifndef HK_REFILL_MIN_IDLE
#define HK_REFILL_MIN_IDLE 8
#endif
// the tracking loops wait on dependent loads (NanoVDB tree, majorant cells) ~60 % of the time at 2 waves/SIMD: trade a few
// spilled registers for a third wave
#ifndef HK_MEDIA_WAVES
#define HK_MEDIA_WAVES 2
#endif
#ifndef HK_TRACK_ADVANCE
#define HK_TRACK_ADVANCE 4
#endif
#ifndef HK_DELTA_ADVANCE
#define HK_DELTA_ADVANCE 6   // k_track's own advance-loop length (the shadow walk keeps HK_TRACK_ADVANCE)
#endif
#ifndef HK_SKIP_ZERO
#define HK_SKIP_ZERO 0
#endif
#ifndef HK_GREY_TRACK_WAVES
#define HK_GREY_TRACK_WAVES 4
#endif
enum { TR_BUSY = -101, TR_EMPTY = -100, TR_SCATTER = -3, TR_ESCAPED = -2 };  // >= 0: reached its surface hit of that material kind

// One open output segment of k_track: the wave owns its count words while it is open.
struct TrackSeg {
    int gw;                       // segment index, -1 = slot free
    int esc, sca;                 // entries in the segment's escaped / scatter queues
    int kind_count[HK_MAX_KINDS]; // entries in its per-kind queues
};
// GREY (scenes whose media are all Grid / NanoVDB with flat sigma_a and sigma_s spectra — DScene::all_grey; the BOMEX example's cloud is
// RGBSpectrum(0) / RGBSpectrum(1)): every ratio the tracker multiplies into beta and r_u is then a spectrum of four EQUAL components
// divided by its own first component, i.e. 1 up to rounding — the general code multiplies by (1 +- 3 ulp) per collision, this
// instantiation leaves beta and r_u alone (they stay in memory, not in registers), carries r_l's rescaling as one float, needs no
// wavelengths and no exp at a cell boundary.  Every DECISION (free-flight distance, absorb / scatter / null, termination) is taken on
// the same first-component arithmetic as in the general code, so the collision sequence of a path is identical.
template <int MM, bool GREY = false>
__global__ void __attribute__((amdgpu_flat_work_group_size(256, 256), amdgpu_waves_per_eu(GREY ? HK_GREY_TRACK_WAVES : HK_MEDIA_WAVES))) k_track(DPathState st, DScene sc, DTables T, DFrame fr, int depth, DStats* stats, const DMedium* __restrict__ media) {
    // `media` == sc.media: a restrict-qualified kernel argument, so the record of a wave-uniform medium index is read with scalar loads
    const int lane = lane_id();
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    unsigned n_coll = 0, n_dda = 0;
    const DPathGen g = st.gen[depth & 1];   // throughput / scattering vertex are updated IN PLACE in the current generation
    const bool ones = depth == 0 && fr.implicit_ones;
    // The wave streams segment after segment (SegStream) WITHOUT draining its lanes in between: the collision count per path is so
    // uneven that a third of the lane-slots of a segment-at-a-time walk sat empty in the segment's tail.  A finished path must be
    // pushed into the queues of ITS segment, so two segments are open at any time — `cur` (tag cur_tag) feeds the refill, the other
    // slot holds the previous segment until its last lane has finished — and every lane carries the tag of its segment.
    SegStream stream = stream_open(st, ticket_ptr(st, depth, TK_TRACK), true, depth, Q_MEDIUM);
    TrackSeg seg[2];
    seg[0].gw = seg[1].gw = -1;
    int cur_tag = 0;
    const uint32_t* __restrict__ queue = st.medium_q;
    int n = 0, cursor = 0;
    bool more = true;
    auto open_seg = [&](TrackSeg& s, int gw) {
        s.gw = gw;
        s.esc = *count_ptr(st, depth, Q_ESCAPED, gw);
        s.sca = 0;
#pragma unroll
        for (int k = 0; k < HK_MAX_KINDS; ++k) s.kind_count[k] = *count_ptr(st, depth, Q_MAT0 + k, gw);
    };
    auto close_seg = [&](TrackSeg& s) {
        if (s.gw >= 0 && lane == 0) {
            *count_ptr(st, depth, Q_SCATTER, s.gw) = s.sca;
            *count_ptr(st, depth, Q_ESCAPED, s.gw) = s.esc;
#pragma unroll
            for (int k = 0; k < HK_MAX_KINDS; ++k) *count_ptr(st, depth, Q_MAT0 + k, s.gw) = s.kind_count[k];
        }
        s.gw = -1;
    };
    {
        int state = TR_EMPTY, tag = 0;
        uint32_t slot = 0, pslot = 0;   // generation index of the lane's path; its path slot (pixel-sample id: where L lives)
        v3 ro = mk3(0, 0, 0), rd = mk3(0, 0, 1), cur_o = mk3(0, 0, 0);
        S4 lambda = s4(0.0f), beta = s4(0.0f), r_u = s4(0.0f), r_l = s4(0.0f), base_a = s4(0.0f), base_s = s4(0.0f), base_Le = s4(0.0f), sm = s4(0.0f);
        float a0 = 0.0f, s0 = 0.0f, rl_f = 1.0f;   // GREY: the flat sigma_a / sigma_s values, the factor r_l has picked up so far
        bool dead_null = false;                    // GREY: beta or r_u is black (the general code notices at the first null collision / at the end)
        uint64_t rng = 0;
        MajorantIter it = exhausted_iter();
        float seg1 = 0.0f, sm0 = 0.0f, t = 0.0f;
        bool in_seg = false, pending = false;
        float pend_dt = 0.0f;
        int k_in_seg = 0, segi = 0, medium_idx = 0;
        for (;;) {
            const unsigned long long busy_m = __ballot(state == TR_BUSY);
            if (busy_m == 0ull || (64 - __popcll(busy_m) >= fr.refill_idle && (cursor < n || more))) {
                // ---- route the finished paths into the queues of their own segment ----
#pragma unroll
                for (int tg = 0; tg < 2; ++tg) {
                    TrackSeg& S = seg[tg];
                    const bool sel = tag == tg && state != TR_BUSY && state != TR_EMPTY;
                    if (__ballot(sel) == 0ull) continue;
                    const size_t base = (size_t)S.gw * st.wave_cap;
                    {
                        const unsigned long long m = __ballot(sel && state == TR_SCATTER);
                        if (sel && state == TR_SCATTER) st.scatter_q[base + S.sca + __popcll(m & lt_mask)] = slot;
                        S.sca += __popcll(m);
                    }
                    {
                        const unsigned long long m = __ballot(sel && state == TR_ESCAPED);
                        if (sel && state == TR_ESCAPED) st.escaped_q[base + S.esc + __popcll(m & lt_mask)] = slot;
                        S.esc += __popcll(m);
                    }
                    unsigned long long todo_k = __ballot(sel && state >= 0);
                    while (todo_k) {
                        int src = __ffsll((long long)todo_k) - 1;
                        const int k = __builtin_amdgcn_readlane(state, src);   // wave-uniform: counts stay in scalar registers
                        bool mine = sel && state == k;
                        unsigned long long m = __ballot(mine);
                        int cnt = 0;
#pragma unroll
                        for (int kk = 0; kk < HK_MAX_KINDS; ++kk) cnt = (kk == k) ? S.kind_count[kk] : cnt;
                        if (mine) st.mat_q[((size_t)k * st.n_waves + S.gw) * st.wave_cap + cnt + __popcll(m & lt_mask)] = slot;
                        int add = __popcll(m);
#pragma unroll
                        for (int kk = 0; kk < HK_MAX_KINDS; ++kk) S.kind_count[kk] += (kk == k) ? add : 0;
                        todo_k &= ~m;
                    }
                }
                if (state != TR_BUSY) state = TR_EMPTY;
                // ---- the current segment is used up: open the next one in the other slot as soon as that slot's last lane is done ----
                while (more && cursor >= n) {
                    const int other = cur_tag ^ 1;
                    const bool other_busy = __ballot(state == TR_BUSY && tag == other) != 0ull;
                    if (other_busy) break;   // both slots hold lanes in flight: no third segment, the refill waits
                    if (other == 0) close_seg(seg[0]); else close_seg(seg[1]);
                    const int gw = stream_next(stream, st.n_waves);
                    if (gw >= st.n_waves) {
                        more = false;
                        break;
                    }
                    if (other == 0) open_seg(seg[0], gw); else open_seg(seg[1], gw);
                    cur_tag = other;
                    queue = st.medium_q + (size_t)gw * st.wave_cap;
                    n = *count_ptr(st, depth, Q_MEDIUM, gw);
                    cursor = 0;
                }
                // ---- refill idle lanes from the current segment's queue ----
                const unsigned long long want = __ballot(state == TR_EMPTY);
                const int avail = n - cursor;
                const int rank = __popcll(want & lt_mask);
                if (state == TR_EMPTY && rank < avail) {
                    slot = queue[cursor + rank];
                    tag = cur_tag;
                    float4 O = g.ray_o[slot], D = g.ray_d[slot];
                    ro = mk3(O.x, O.y, O.z);
                    rd = mk3(D.x, D.y, D.z);
                    const float t_max = st.hit[slot].x;
                    const uint2 meta = g.meta[slot];
                    medium_idx = (int)(meta.x >> 16) - 1;
                    pslot = meta.y;
                    const DMedium& med = sc.n_media == 1 ? media[0] : media[medium_idx];
                    if constexpr (GREY) {
                        a0 = eval_flat(med.sigma_a);
                        s0 = eval_flat(med.sigma_s);
                        base_a = s4(a0);
                        base_s = s4(s0);
                        rl_f = 1.0f;
                        dead_null = is_black(ld_throughput(g.beta, slot, ones)) || is_black(ld_throughput(g.r_u, slot, ones));
                    } else {
                        lambda = ld_lambda(st, g, slot, pslot);
                        beta = ld_throughput(g.beta, slot, ones);
                        r_u = ld_throughput(g.r_u, slot, ones);
                        r_l = ld_throughput(g.r_l, slot, ones);
                        base_a = eval_scaled(med.sigma_a, lambda);
                        base_s = eval_scaled(med.sigma_s, lambda);
                        if (HK_HAS_MEDIUM(MM, HK_MEDIUM_HOMOGENEOUS)) base_Le = eval_scaled(med.Le, lambda);
                    }
                    rng = lcg_init(ro, rd, t_max);
                    it = create_majorant_iterator<MM>(med, ro, rd, t_max);
                    in_seg = false;
                    pending = false;
                    segi = 0;
                    state = TR_BUSY;
                }
                const int want_n = __popcll(want);
                cursor += want_n < avail ? want_n : (avail > 0 ? avail : 0);
                if (__ballot(state == TR_BUSY) == 0ull) {
                    if (cursor >= n && !more) break;
                    continue;
                }
            }
            // The medium record is read through a wave-uniform index (scalar loads into SGPRs): lanes are served medium by medium
            // ("waterfall"), which is a single trip for the usual one-medium scene.
            bool survived = false;
            unsigned long long todo = __ballot(state == TR_BUSY);
            while (todo) {
                const int m_uniform = __builtin_amdgcn_readlane(medium_idx, __ffsll((long long)todo) - 1);
                const bool mine = state == TR_BUSY && medium_idx == m_uniform;
                const DMedium& med = media[m_uniform];
                // ---- phase A: cheap steps (next majorant cell, free-flight sample, cell-boundary crossing) until every busy lane
                //      either holds a tentative collision or has run out of segments ----
    #pragma unroll 1
                for (int adv = 0; adv < fr.delta_advance; ++adv) {
                    const bool need = mine && state == TR_BUSY && !pending && !survived;
                    if (__ballot(need) == 0ull) break;
                    if (!need) continue;
                    if (!in_seg) {
                        float seg0;
                        if (segi >= 256 || !majorant_next<MM>(it, med, base_a + base_s, seg0, seg1, sm))
                            survived = true;  // ran out of segments with the path still alive
                        else {
                            ++segi;
                            ++n_dda;
                            sm0 = sm.x;
                            if (sm0 >= 1e-10f) {
                                t = seg0;
                                cur_o = ro + rd * t;
                                in_seg = true;
                                k_in_seg = 0;
                            }
                        }
                    }
                    // a lane that has just entered a cell draws its first free flight in the same round (same operations per path
                    // in the same order; one round less per cell entered)
                    if (!in_seg || survived) {
                    } else if (k_in_seg >= 1024) {
                        in_seg = false;
                    } else {
                        ++k_in_seg;
                        float u = lcg_next(rng);
                        pend_dt = -media_logf(maxf(1e-10f, 1.0f - u)) / sm0;
                        float ts = t + pend_dt;
                        if (ts >= seg1) {
                            if constexpr (!GREY) {   // GREY: T_maj / T_maj[1] = 1, nothing changes
                                float dr = seg1 - t;
                                S4 Tm = s4exp((-dr) * sm);
                                float T0 = Tm.x;
                                if (T0 > 1e-10f) {
                                    beta = div4(beta * Tm, T0);
                                    r_u = div4(r_u * Tm, T0);
                                    r_l = div4(r_l * Tm, T0);
                                }
                            }
                            in_seg = false;
                        } else
                            pending = true;
                    }
                }
                // ---- phase B: the tentative collisions (medium lookup, absorb / scatter / null) ----
                if (GREY && mine && state == TR_BUSY && pending) {
                    pending = false;
                    const float dt = pend_dt;
                    const float Tm0 = media_expf((-dt) * sm0);
                    const v3 p = cur_o + rd * dt;
                    ++n_coll;
                    const float d = sample_density<MM>(med, p);
                    const float sa = a0 * d, ss = s0 * d;
                    const float p_absorb = sa / sm0, p_scatter = ss / sm0;
                    const float ue = lcg_next(rng);
                    if (ue < p_absorb) {
                        state = TR_EMPTY;  // absorbed
                    } else if (ue < p_absorb + p_scatter) {
                        if (depth >= fr.max_depth)
                            state = TR_EMPTY;
                        else {   // beta and r_u are rescaled by sigma_s T_maj / (sigma_s T_maj)[1] = 1: they stay as they are in the record
                            g.ray_o[slot] = make_float4(p.x, p.y, p.z, INF_F);  // the scattering vertex: k_scatter continues from here
                            state = TR_SCATTER;
                        }
                    } else {
                        const float sn0 = maxf(sm0 - sa - ss, 0.0f);
                        const float pdf = Tm0 * sn0;
                        if (pdf > 1e-10f) {
                            rl_f = ((rl_f * Tm0) * sm0) * (1.0f / pdf);
                            t = t + dt;
                            cur_o = p;
                            if (dead_null) state = TR_EMPTY;
                        } else
                            state = TR_EMPTY;
                    }
                }
                if (!GREY && mine && state == TR_BUSY && pending) {
                    pending = false;
                    const float dt = pend_dt;
                    const float ts = t + dt;
                    S4 Tm = s4exp((-dt) * sm);
                    v3 p = cur_o + rd * dt;
                    ++n_coll;
                    MediumProps mp = sample_point<MM>(T, lambda, med, base_a, base_s, base_Le, p);
                    if ((HK_HAS_MEDIUM(MM, HK_MEDIUM_HOMOGENEOUS) || HK_HAS_MEDIUM(MM, HK_MEDIUM_RGB_GRID)) && !is_black(mp.Le) && depth < fr.max_depth) {
                        float pr = sm0 * Tm.x;
                        if (pr > 1e-10f) {
                            S4 r_e = r_u * sm * Tm / pr;
                            if (!is_black(r_e)) st4(&st.L[pslot], ld4(&st.L[pslot]) + beta * mp.sigma_a * Tm * mp.Le / (pr * average(r_e)));
                        }
                    }
                    float p_absorb = mp.sigma_a.x / sm0, p_scatter = mp.sigma_s.x / sm0;
                    float ue = lcg_next(rng);
                    if (ue < p_absorb) {
                        state = TR_EMPTY;  // absorbed
                    } else if (ue < p_absorb + p_scatter) {
                        if (depth >= fr.max_depth)
                            state = TR_EMPTY;
                        else {
                            float pdf = Tm.x * mp.sigma_s.x;
                            if (pdf > 1e-10f) {
                                beta = div4(beta * Tm * mp.sigma_s, pdf);
                                r_u = div4(r_u * Tm * mp.sigma_s, pdf);
                            }
                            st4(&g.beta[slot], beta);
                            st4(&g.r_u[slot], r_u);
                            g.ray_o[slot] = make_float4(p.x, p.y, p.z, INF_F);  // the scattering vertex: k_scatter continues from here
                            state = TR_SCATTER;
                        }
                    } else {
                        S4 sn = s4max0(sm - mp.sigma_a - mp.sigma_s);
                        float pdf = Tm.x * sn.x;
                        if (pdf > 1e-10f) {
                            beta = div4(beta * Tm * sn, pdf);
                            r_u = div4(r_u * Tm * sn, pdf);
                            r_l = div4(r_l * Tm * sm, pdf);
                            t = ts;
                            cur_o = p;
                            if (is_black(beta) || is_black(r_u)) state = TR_EMPTY;
                        } else
                            state = TR_EMPTY;
                    }
                }
                todo &= ~__ballot(mine);
            }
            if (survived) {
                state = TR_EMPTY;
                if (!((GREY ? dead_null : (is_black(beta) || is_black(r_u))) || depth >= fr.max_depth)) {
                    // survived to t_max: hand the stored surface hit / the escape over with the updated throughput
                    if constexpr (GREY) {
                        st4(&g.r_l[slot], ld_throughput(g.r_l, slot, ones) * rl_f);
                    } else {
                        st4(&g.beta[slot], beta);
                        st4(&g.r_u[slot], r_u);
                        st4(&g.r_l[slot], r_l);
                    }
                    float4 H = st.hit[slot];
                    const int prim = __float_as_int(H.y);
                    if (prim < 0)
                        state = TR_ESCAPED;
                    else {
                        const int mat_word = st.mat_id[slot];
                        int mat = mat_word & ~HK_MAT_EMISSIVE_BIT;
                        if (sc.materials[mat].kind == HK_MAT_MIX) {
                            float w = 1.0f - H.z - H.w;
                            mat = resolve_mix_material(sc, mat, ro + rd * H.x, -rd, uv_at(sc, prim, w, H.z, H.w));
                            st.mat_id[slot] = mat | (mat_word & HK_MAT_EMISSIVE_BIT);
                        }
                        int kind = sc.materials[mat].kind;
                        state = kind == HK_MAT_MIX ? HK_MAT_FALLBACK : kind;
                    }
                }
            }
        }
    }
    close_seg(seg[0]);
    close_seg(seg[1]);
    stats += global_wave();
    wave_add(&stats->collisions, n_coll);
    wave_add(&stats->nvdb_collisions, (sc.media_mask >> HK_MEDIUM_NANOVDB) & 1 ? n_coll : 0u);   // (priced per scene: SURVEY 8d charges the NanoVDB tree walk)
    wave_add(&stats->dda_steps, n_dda);
}

// ---------------------------------------------------------------------------------------------------
// k_track_flat: K4 of a scene whose ONE medium is GREY (k_track<MM, true> above) — the same per-lane state machine, the same
// arithmetic and RNG consumption per path (films bit-identical: tests + tools/ab_bitwise.py, HK_GREY_FLAT=0 runs the older kernel),
// re-shaped for INSTRUCTION ISSUE.  On this chip a scalar instruction costs a SIMD about what a vector one does (tools/valu_rate.hip:
// v_fma + s_add pair 6.4 cycles at 4 waves against 2.9 + 4.2 alone), and k_track issued as many scalar exec-mask instructions as
// vector ones: every loop-carried `bool` (in_seg, pending, survived, dead_null, the iterator's liveness ...) lived in an SGPR pair
// and every divergent assignment to it cost an andn2 / and / or triple PER NESTING LEVEL it crossed.  Here
//   * the lane's flags are bits of ONE VGPR: a write under a divergent exec mask merges for free, a test is v_and + v_cmp;
//   * the majorant iterator's mode / axis signs are three more bits of that word, its DDA step is straight-line code;
//   * there is one medium (all_grey): no waterfall loop, its record's fields are read once into scalar registers before the loop;
//   * BRICKS: the medium is a NanoVDB grid with dense halo bricks (DScene::grey_bricks) — the tree-walk paths of the lookup are
//     not in the kernel;
//   * the collision round can WAIT (fr.track_gate): when fewer than N lanes hold a tentative collision and others can still advance,
//     the cheap-step loop goes on for a few more iterations (the rule of k_shadow's waiting leaf phases).
// ---------------------------------------------------------------------------------------------------
#ifndef HK_FLAT_TRACK_WAVES
#define HK_FLAT_TRACK_WAVES 4
#endif
enum { TF_IN_SEG = 1, TF_PENDING = 2, TF_SURVIVED = 4, TF_DEAD_NULL = 8, TF_TAG = 16, TF_IT_LIVE = 32, TF_NEG0 = 0x100, TF_NEG1 = 0x200, TF_NEG2 = 0x400 };
template <int MM, bool BRICKS>
__global__ void __attribute__((amdgpu_flat_work_group_size(256, 256), amdgpu_waves_per_eu(HK_FLAT_TRACK_WAVES))) k_track_flat(DPathState st, DScene sc, DFrame fr, int depth, DStats* stats, const DMedium* __restrict__ media) {
    const int lane = lane_id();
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    unsigned n_coll = 0, n_dda = 0;
    HK_DBG_DECL
    const DPathGen g = st.gen[depth & 1];
    const bool ones = depth == 0 && fr.implicit_ones;
    const DMedium& med = media[0];
    const float a0 = eval_flat(med.sigma_a), s0 = eval_flat(med.sigma_s);
    const float sig_t = a0 + s0;   // (base_a + base_s).x of the general code
    const int mrx = med.mres[0], mry = med.mres[1], mrz = med.mres[2];
    const float* __restrict__ maj = med.majorant;
    const bool last_depth = depth >= fr.max_depth;
    const int gate_min = fr.track_gate & 0xff, gate_cap = fr.delta_advance + ((fr.track_gate >> 8) & 0xff);
    SegStream stream = stream_open(st, ticket_ptr(st, depth, TK_TRACK), true, depth, Q_MEDIUM);
    TrackSeg seg[2];
    seg[0].gw = seg[1].gw = -1;
    int cur_tag = 0;
    const uint32_t* __restrict__ queue = st.medium_q;
    int n = 0, cursor = 0;
    bool more = true;
    auto open_seg = [&](TrackSeg& s, int gw) {
        s.gw = gw;
        s.esc = *count_ptr(st, depth, Q_ESCAPED, gw);
        s.sca = 0;
#pragma unroll
        for (int k = 0; k < HK_MAX_KINDS; ++k) s.kind_count[k] = *count_ptr(st, depth, Q_MAT0 + k, gw);
    };
    auto close_seg = [&](TrackSeg& s) {
        if (s.gw >= 0 && lane == 0) {
            *count_ptr(st, depth, Q_SCATTER, s.gw) = s.sca;
            *count_ptr(st, depth, Q_ESCAPED, s.gw) = s.esc;
#pragma unroll
            for (int k = 0; k < HK_MAX_KINDS; ++k) *count_ptr(st, depth, Q_MAT0 + k, s.gw) = s.kind_count[k];
        }
        s.gw = -1;
    };
    int state = TR_EMPTY, fl = 0;
    uint32_t slot = 0;
    v3 ro = mk3(0, 0, 0), rd = mk3(0, 0, 1), cur_o = mk3(0, 0, 0);
    float rl_f = 1.0f;
    uint64_t rng = 0;
    float it_tmin = 0.0f, it_tmax = 0.0f, nt0 = 0.0f, nt1 = 0.0f, nt2 = 0.0f, dl0 = 0.0f, dl1 = 0.0f, dl2 = 0.0f;
    int vx = 0, vy = 0, vz = 0;
    float seg1 = 0.0f, sm0 = 0.0f, t = 0.0f, pend_dt = 0.0f;
    int k_in_seg = 0, segi = 0;
    for (;;) {
        const unsigned long long busy_m = __ballot(state == TR_BUSY);
        if (busy_m == 0ull || (64 - __popcll(busy_m) >= fr.refill_idle && (cursor < n || more))) {
            // ---- route the finished paths into the queues of their own segment ----
            const int tag = (fl & TF_TAG) ? 1 : 0;
#pragma unroll
            for (int tg = 0; tg < 2; ++tg) {
                TrackSeg& S = seg[tg];
                const bool sel = tag == tg && state != TR_BUSY && state != TR_EMPTY;
                if (__ballot(sel) == 0ull) continue;
                const size_t base = (size_t)S.gw * st.wave_cap;
                {
                    const unsigned long long m = __ballot(sel && state == TR_SCATTER);
                    if (sel && state == TR_SCATTER) st.scatter_q[base + S.sca + __popcll(m & lt_mask)] = slot;
                    S.sca += __popcll(m);
                }
                {
                    const unsigned long long m = __ballot(sel && state == TR_ESCAPED);
                    if (sel && state == TR_ESCAPED) st.escaped_q[base + S.esc + __popcll(m & lt_mask)] = slot;
                    S.esc += __popcll(m);
                }
                unsigned long long todo_k = __ballot(sel && state >= 0);
                while (todo_k) {
                    int src = __ffsll((long long)todo_k) - 1;
                    const int k = __builtin_amdgcn_readlane(state, src);   // wave-uniform: counts stay in scalar registers
                    bool mine = sel && state == k;
                    unsigned long long m = __ballot(mine);
                    int cnt = 0;
#pragma unroll
                    for (int kk = 0; kk < HK_MAX_KINDS; ++kk) cnt = (kk == k) ? S.kind_count[kk] : cnt;
                    if (mine) st.mat_q[((size_t)k * st.n_waves + S.gw) * st.wave_cap + cnt + __popcll(m & lt_mask)] = slot;
                    int add = __popcll(m);
#pragma unroll
                    for (int kk = 0; kk < HK_MAX_KINDS; ++kk) S.kind_count[kk] += (kk == k) ? add : 0;
                    todo_k &= ~m;
                }
            }
            if (state != TR_BUSY) state = TR_EMPTY;
            // ---- the current segment is used up: open the next one in the other slot as soon as that slot's last lane is done ----
            while (more && cursor >= n) {
                const int other = cur_tag ^ 1;
                const bool other_busy = __ballot(state == TR_BUSY && ((fl & TF_TAG) ? 1 : 0) == other) != 0ull;
                if (other_busy) break;   // both slots hold lanes in flight: no third segment, the refill waits
                if (other == 0) close_seg(seg[0]); else close_seg(seg[1]);
                const int gw = stream_next(stream, st.n_waves);
                if (gw >= st.n_waves) {
                    more = false;
                    break;
                }
                if (other == 0) open_seg(seg[0], gw); else open_seg(seg[1], gw);
                cur_tag = other;
                queue = st.medium_q + (size_t)gw * st.wave_cap;
                n = *count_ptr(st, depth, Q_MEDIUM, gw);
                cursor = 0;
            }
            // ---- refill idle lanes from the current segment's queue ----
            const unsigned long long want = __ballot(state == TR_EMPTY);
            const int avail = n - cursor;
            const int rank = __popcll(want & lt_mask);
            HK_DBG(5, state == TR_EMPTY && rank < avail);
            if (state == TR_EMPTY && rank < avail) {
                slot = queue[cursor + rank];
                float4 O = g.ray_o[slot], D = g.ray_d[slot];
                ro = mk3(O.x, O.y, O.z);
                rd = mk3(D.x, D.y, D.z);
                const float t_max = st.hit[slot].x;
                const bool dead = is_black(ld_throughput(g.beta, slot, ones)) || is_black(ld_throughput(g.r_u, slot, ones));
                rl_f = 1.0f;
                rng = lcg_init(ro, rd, t_max);
                const MajorantIter it = create_majorant_iterator<MM>(med, ro, rd, t_max);
                it_tmin = it.t_min;
                it_tmax = it.t_max;
                nt0 = it.next_t[0], nt1 = it.next_t[1], nt2 = it.next_t[2];
                dl0 = it.delta_t[0], dl1 = it.delta_t[1], dl2 = it.delta_t[2];
                vx = it.voxel[0], vy = it.voxel[1], vz = it.voxel[2];
                fl = (cur_tag ? TF_TAG : 0) | (dead ? TF_DEAD_NULL : 0) | ((it.mode & 0xff) == 2 ? TF_IT_LIVE : 0) | (it.mode & 0x700);
                segi = 0;
                state = TR_BUSY;
            }
            const int want_n = __popcll(want);
            cursor += want_n < avail ? want_n : (avail > 0 ? avail : 0);
            if (__ballot(state == TR_BUSY) == 0ull) {
                if (cursor >= n && !more) break;
                continue;
            }
        }
        // ---- phase A: cheap steps (next majorant cell, free-flight sample) until the busy lanes hold a tentative collision or have
        //      run out of cells.  fr.delta_advance iterations; then, while fewer than gate_min lanes hold a collision, up to gate_cap ----
#pragma unroll 1
        for (int adv = 0;; ++adv) {
            const bool need = state == TR_BUSY && (fl & (TF_PENDING | TF_SURVIVED)) == 0;
            if (__ballot(need) == 0ull) break;
            if (adv >= fr.delta_advance) {
                if (adv >= gate_cap) break;
                if (__popcll(__ballot(state == TR_BUSY && (fl & TF_PENDING) != 0)) >= gate_min) break;
            }
            HK_DBG(0, need);
            if (need) {
                if ((fl & TF_IN_SEG) == 0) {
                    HK_DBG(1, true);
                    // majorant_next (media.jl:625-729) of a DDA iterator, straight-line
                    if ((fl & TF_IT_LIVE) == 0 || segi >= 256 || it_tmin >= it_tmax)
                        fl |= TF_SURVIVED;   // ran out of segments with the path still alive
                    else {
                        const bool lxy = nt0 < nt1, lxz = nt0 < nt2, lyz = nt1 < nt2;
                        const bool ax0 = lxy & lxz, ax1 = (!lxy) & lyz;   // axis 0, axis 1, else axis 2
                        const float nt = ax0 ? nt0 : (ax1 ? nt1 : nt2);
                        const float stm = minf(nt, it_tmax);
                        const float rho = maj[vx + mrx * (vy + mry * vz)];
                        const float seg0 = it_tmin;
                        seg1 = stm;
                        const bool neg = (fl & (ax0 ? TF_NEG0 : (ax1 ? TF_NEG1 : TF_NEG2))) != 0;
                        const int v = (ax0 ? vx : (ax1 ? vy : vz)) + (neg ? -1 : 1);
                        const int lim = neg ? -1 : (ax0 ? mrx : (ax1 ? mry : mrz));
                        const float s = nt + (ax0 ? dl0 : (ax1 ? dl1 : dl2));
                        vx = ax0 ? v : vx;
                        vy = ax1 ? v : vy;
                        vz = (ax0 | ax1) ? vz : v;
                        nt0 = ax0 ? s : nt0;
                        nt1 = ax1 ? s : nt1;
                        nt2 = (ax0 | ax1) ? nt2 : s;
                        const bool out = v == lim;
                        fl = out ? (fl & ~TF_IT_LIVE) : fl;
                        it_tmin = out ? it_tmax : stm;
                        ++segi;
                        ++n_dda;
                        sm0 = sig_t * rho;
                        if (sm0 >= 1e-10f) {
                            t = seg0;
                            cur_o = ro + rd * t;
                            fl |= TF_IN_SEG;
                            k_in_seg = 0;
                        }
                    }
                }
                // a lane that has just entered a cell draws its first free flight in the same round
                if ((fl & (TF_IN_SEG | TF_SURVIVED)) == TF_IN_SEG) {
                    HK_DBG(2, true);
                    if (k_in_seg >= 1024)
                        fl &= ~TF_IN_SEG;
                    else {
                        ++k_in_seg;
                        const float u = lcg_next(rng);
                        pend_dt = -media_logf(maxf(1e-10f, 1.0f - u)) / sm0;
                        const float ts = t + pend_dt;
                        fl = ts >= seg1 ? (fl & ~TF_IN_SEG) : (fl | TF_PENDING);   // leaves the cell (T_maj / T_maj[1] = 1: nothing else changes) / tentative collision
                    }
                }
            }
        }
        // ---- phase B: the tentative collisions (medium lookup, absorb / scatter / null) ----
        HK_DBG(3, state == TR_BUSY && (fl & TF_PENDING) != 0);
        HK_DBG(4, state == TR_BUSY);
        if (state == TR_BUSY && (fl & TF_PENDING) != 0) {
            fl &= ~TF_PENDING;
            const float dt = pend_dt;
            const float Tm0 = media_expf((-dt) * sm0);
            const v3 p = cur_o + rd * dt;
            ++n_coll;
            const float d = sample_density<MM, BRICKS>(med, p);
            const float sa = a0 * d, ss = s0 * d;
            const float p_absorb = sa / sm0, p_scatter = ss / sm0;
            const float ue = lcg_next(rng);
            if (ue < p_absorb) {
                state = TR_EMPTY;  // absorbed
            } else if (ue < p_absorb + p_scatter) {
                if (last_depth)
                    state = TR_EMPTY;
                else {   // beta and r_u are rescaled by sigma_s T_maj / (sigma_s T_maj)[1] = 1: they stay as they are in the record
                    g.ray_o[slot] = make_float4(p.x, p.y, p.z, INF_F);  // the scattering vertex: k_scatter continues from here
                    state = TR_SCATTER;
                }
            } else {
                const float sn0 = maxf(sm0 - sa - ss, 0.0f);
                const float pdf = Tm0 * sn0;
                if (pdf > 1e-10f) {
                    rl_f = ((rl_f * Tm0) * sm0) * (1.0f / pdf);
                    t = t + dt;
                    cur_o = p;
                    if (fl & TF_DEAD_NULL) state = TR_EMPTY;
                } else
                    state = TR_EMPTY;
            }
        }
        if (state == TR_BUSY && (fl & TF_SURVIVED) != 0) {
            state = TR_EMPTY;
            if (!((fl & TF_DEAD_NULL) != 0 || last_depth)) {
                // survived to t_max: hand the stored surface hit / the escape over with the updated throughput
                st4(&g.r_l[slot], ld_throughput(g.r_l, slot, ones) * rl_f);
                float4 H = st.hit[slot];
                const int prim = __float_as_int(H.y);
                if (prim < 0)
                    state = TR_ESCAPED;
                else {
                    const int mat_word = st.mat_id[slot];
                    int mat = mat_word & ~HK_MAT_EMISSIVE_BIT;
                    if (sc.materials[mat].kind == HK_MAT_MIX) {
                        float w = 1.0f - H.z - H.w;
                        mat = resolve_mix_material(sc, mat, ro + rd * H.x, -rd, uv_at(sc, prim, w, H.z, H.w));
                        st.mat_id[slot] = mat | (mat_word & HK_MAT_EMISSIVE_BIT);
                    }
                    int kind = sc.materials[mat].kind;
                    state = kind == HK_MAT_MIX ? HK_MAT_FALLBACK : kind;
                }
            }
        }
    }
    close_seg(seg[0]);
    close_seg(seg[1]);
    stats += global_wave();
    HK_DBG_FLUSH(stats);
    wave_add(&stats->collisions, n_coll);
    wave_add(&stats->nvdb_collisions, (sc.media_mask >> HK_MEDIUM_NANOVDB) & 1 ? n_coll : 0u);   // (priced per scene: SURVEY 8d charges the NanoVDB tree walk)
    wave_add(&stats->dda_steps, n_dda);
}

// ---------------------------------------------------------------------------------------------------
// k_track_pool: k_track_flat with DENSE set-up and DENSE routing through a per-wave pool in LDS.  A camera / bounce ray lives 2.7
// collisions and 7.7 majorant cells, so half of k_track's instructions were the per-ray set-up (queue entry, ray record, LCG seed,
// majorant iterator: 18 IEEE divisions) and the routing of the finished paths — both executed for the ~25 idle lanes that triggered a
// refill, with every instruction paid for 64.  Here
//   * SET-UP runs for 64 queue entries at once, whenever the pool is empty and a lane is free, and writes 64 ready-to-track states
//     (19 words each) to the wave's pool;
//   * a lane whose path ends takes the next state from the pool IN THE SAME ROUND (19 LDS reads) — no lane waits for a refill round;
//   * a finished path leaves (slot, destination key, r_l factor) in a 128-entry ring in LDS; the ring is ROUTED 64 entries at a time:
//     the r_l update, the hit / material classification of the survivors (Mix resolve included) and the ballot compaction into the
//     segment's queues all run on full waves;
//   * the two open segments' count words live in LDS, not in 28 scalar registers.
// Same arithmetic and RNG consumption per path as k_track<MM, true> (films bit-identical: tools/ab_bitwise.py; HK_TRACK_POOL=0 runs
// k_track_flat).  Only the ORDER of a segment's queue entries differs, which no result depends on (test_scheduling_is_result_neutral).
// ---------------------------------------------------------------------------------------------------
enum { TP_SLOT = 0, TP_RO = 1, TP_RD = 4, TP_RNG = 7, TP_TMIN = 9, TP_TMAX = 10, TP_NT = 11, TP_DL = 14, TP_VOX = 17, TP_FL = 18, TP_FIELDS = 19, TP_RING = 128,
       TP_TAGS = 4, TP_TAG_SHIFT = 12, TP_WAVE_INTS = TP_FIELDS * 64 + 3 * TP_RING + 16 * TP_TAGS, TP_SURVIVED = 14 };
HKD void wave_lds_fence() {   // orders this wave's LDS writes before its later LDS reads by OTHER lanes (the hardware executes a wave's LDS instructions in order)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
template <int MM, bool BRICKS>
__global__ void __attribute__((amdgpu_flat_work_group_size(256, 256), amdgpu_waves_per_eu(HK_FLAT_TRACK_WAVES))) k_track_pool(DPathState st, DScene sc, DFrame fr, int depth, DStats* stats, const DMedium* __restrict__ media) {
    __shared__ int lds_all[4 * TP_WAVE_INTS];
    int* const pool = lds_all + (threadIdx.x >> 6) * TP_WAVE_INTS;   // [field][64]
    int* const ring = pool + TP_FIELDS * 64;                          // [slot | key | r_l factor][TP_RING]
    int* const segc = ring + 3 * TP_RING;                             // [tag][16]: entries in the segment's escaped, scatter and per-kind queues; [15] = the segment
    const int lane = lane_id();
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    unsigned n_coll = 0, n_dda = 0;
    HK_DBG_DECL
    const DPathGen g = st.gen[depth & 1];
    const bool ones = depth == 0 && fr.implicit_ones;
    const DMedium& med = media[0];
    const float a0 = eval_flat(med.sigma_a), s0 = eval_flat(med.sigma_s);
    const float sig_t = a0 + s0;   // (base_a + base_s).x of the general code
    const int mrx = med.mres[0], mry = med.mres[1], mrz = med.mres[2];
    const float* __restrict__ maj = med.majorant;
    const bool last_depth = depth >= fr.max_depth;
    const int gate_min = fr.track_gate & 0xff, gate_cap = fr.delta_advance + ((fr.track_gate >> 8) & 0xff);
    SegStream stream = stream_open(st, ticket_ptr(st, depth, TK_TRACK), true, depth, Q_MEDIUM);
    int cur_tag = 0;          // up to TP_TAGS segments are open at a time (a path's tag: bits TP_TAG_SHIFT.. of its flag word): the wave goes on
                              // with the next segment while the last paths of the previous ones are still in flight
    if (lane < TP_TAGS) segc[lane * 16 + 15] = -1;
    const uint32_t* __restrict__ queue = st.medium_q;
    int n = 0, cursor = 0;
    bool more = true;
    int pool_n = 0, res_head = 0, res_n = 0;   // wave-uniform
    bool drain = false;                         // route the whole ring, not only full chunks
    auto open_seg = [&](int tag, int gw) {
        if (lane < 2 + HK_MAX_KINDS) segc[tag * 16 + lane] = lane == 1 ? 0 : *count_ptr(st, depth, lane == 0 ? Q_ESCAPED : Q_MAT0 + lane - 2, gw);
        if (lane == 15) segc[tag * 16 + 15] = gw;
    };
    auto close_seg = [&](int tag) {
        const int gw = __builtin_amdgcn_readfirstlane(segc[tag * 16 + 15]);
        if (gw >= 0 && lane < 2 + HK_MAX_KINDS) *count_ptr(st, depth, lane == 0 ? Q_ESCAPED : (lane == 1 ? Q_SCATTER : Q_MAT0 + lane - 2), gw) = segc[tag * 16 + lane];
        if (lane == 15) segc[tag * 16 + 15] = -1;
    };
    // route the first cnt (<= 64) entries of the ring: r_l update and classification of the survivors, then ballot compaction per destination
    auto route = [&](int cnt) {
        wave_lds_fence();
        const bool have = lane < cnt;
        const int p = (res_head + lane) & (TP_RING - 1);
        uint32_t rslot = 0;
        int key = -1;
        if (have) {
            rslot = (uint32_t)ring[p];
            key = ring[TP_RING + p];
        }
        HK_DBG(6, have);
        if (have && (key & 15) == TP_SURVIVED) {
            // survived to t_max: hand the stored surface hit / the escape over with the updated throughput
            const float rl = __int_as_float(ring[2 * TP_RING + p]);
            mul_rl(g, rslot, ones, rl, st.compact != 0);
            const float4 H = st.hit[rslot];
            const int prim = __float_as_int(H.y);
            int cls = 0;
            if (prim >= 0) {
                const int mat_word = st.mat_id[rslot];
                int mat = mat_word & ~HK_MAT_EMISSIVE_BIT;
                if (sc.materials[mat].kind == HK_MAT_MIX) {
                    const float4 O = g.ray_o[rslot], D = g.ray_d[rslot];
                    const v3 o = mk3(O.x, O.y, O.z), d = mk3(D.x, D.y, D.z);
                    const float w = 1.0f - H.z - H.w;
                    mat = resolve_mix_material(sc, mat, o + d * H.x, -d, uv_at(sc, prim, w, H.z, H.w));
                    st.mat_id[rslot] = mat | (mat_word & HK_MAT_EMISSIVE_BIT);
                }
                const int kind = sc.materials[mat].kind;
                cls = 2 + (kind == HK_MAT_MIX ? HK_MAT_FALLBACK : kind);
            }
            key = (key & ~15) | cls;
        }
        res_head = (res_head + cnt) & (TP_RING - 1);
        res_n -= cnt;
        unsigned long long todo = __ballot(have);
        while (todo) {
            const int k = __builtin_amdgcn_readlane(key, __ffsll((long long)todo) - 1);   // wave-uniform destination
            const unsigned long long m = __ballot(key == k);
            const int base = __builtin_amdgcn_readfirstlane(segc[k]);
            const int gw = __builtin_amdgcn_readfirstlane(segc[(k & ~15) + 15]), cls = k & 15;
            uint32_t* __restrict__ q = cls == 0 ? st.escaped_q + (size_t)gw * st.wave_cap
                                     : cls == 1 ? st.scatter_q + (size_t)gw * st.wave_cap
                                                : st.mat_q + ((size_t)(cls - 2) * st.n_waves + gw) * st.wave_cap;
            if (key == k) q[base + __popcll(m & lt_mask)] = rslot;
            if (lane == 0) segc[k] = base + __popcll(m);
            todo &= ~m;
        }
        wave_lds_fence();
    };
    int state = TR_EMPTY, fl = 0;
    uint32_t slot = 0;
    v3 ro = mk3(0, 0, 0), rd = mk3(0, 0, 1), cur_o = mk3(0, 0, 0);
    float rl_f = 1.0f;
    uint64_t rng = 0;
    float it_tmin = 0.0f, it_tmax = 0.0f, nt0 = 0.0f, nt1 = 0.0f, nt2 = 0.0f, dl0 = 0.0f, dl1 = 0.0f, dl2 = 0.0f;
    int vx = 0, vy = 0, vz = 0;
    float seg1 = 0.0f, sm0 = 0.0f, t = 0.0f, pend_dt = 0.0f;
    int k_in_seg = 0, segi = 0;
    for (;;) {
        // ---- finished paths leave their result in the ring ----
        {
            const bool fin = state != TR_BUSY && state != TR_EMPTY;
            const unsigned long long m = __ballot(fin);
            if (m != 0ull) {
                if (fin) {
                    const int p = (res_head + res_n + __popcll(m & lt_mask)) & (TP_RING - 1);
                    ring[p] = (int)slot;
                    ring[TP_RING + p] = (((fl >> TP_TAG_SHIFT) & (TP_TAGS - 1)) << 4) | (state == TR_SCATTER ? 1 : TP_SURVIVED);   // (survivors are classified when routed)
                    ring[2 * TP_RING + p] = __float_as_int(rl_f);
                    state = TR_EMPTY;
                }
                res_n += __popcll(m);
            }
        }
        // full chunks of the ring are routed at once; all of it before a segment is retired and at the end (ONE call site: the routing
        // code with its Mix resolve is long)
        while (res_n >= (drain ? 1 : 64)) route(res_n < 64 ? res_n : 64);
        drain = false;
        unsigned long long free_m = __ballot(state == TR_EMPTY);
        // ---- the pool is empty and a lane is free: set up the next 64 entries of the segment's queue ----
        if (free_m != 0ull && pool_n == 0 && (cursor < n || more)) {
            while (more && cursor >= n) {
                // the current segment is used up: open the next one in the oldest slot as soon as that slot's last lane is done
                const int other = (cur_tag + 1) & (TP_TAGS - 1);
                const bool other_busy = __ballot(state == TR_BUSY && ((fl >> TP_TAG_SHIFT) & (TP_TAGS - 1)) == other) != 0ull;
                if (other_busy) break;   // every slot holds lanes in flight: the set-up waits
                if (res_n > 0) {         // its last results are still in the ring
                    drain = true;
                    break;
                }
                close_seg(other);
                const int gw = stream_next(stream, st.n_waves);
                if (gw >= st.n_waves) {
                    more = false;
                    break;
                }
                open_seg(other, gw);
                cur_tag = other;
                queue = st.medium_q + (size_t)gw * st.wave_cap;
                n = *count_ptr(st, depth, Q_MEDIUM, gw);
                cursor = 0;
            }
            if (drain) continue;
            const int avail = n - cursor;
            const int take = avail < 64 ? avail : 64;
            if (take > 0) HK_DBG(5, lane < take);
            if (lane < take) {
                const uint32_t s_new = queue[cursor + lane];
                const float4 O = g.ray_o[s_new], D = g.ray_d[s_new];
                const v3 o = mk3(O.x, O.y, O.z), d = mk3(D.x, D.y, D.z);
                const float t_max = st.hit[s_new].x;
                const bool dead = is_black(ld_throughput(g.beta, s_new, ones)) || is_black(ld_ru(g, s_new, ones, st.compact != 0));
                const uint64_t r = lcg_init(o, d, t_max);
                const MajorantIter it = create_majorant_iterator<MM>(med, o, d, t_max);
                int* e = pool + lane;
                e[TP_SLOT * 64] = (int)s_new;
                e[(TP_RO + 0) * 64] = __float_as_int(o.x);
                e[(TP_RO + 1) * 64] = __float_as_int(o.y);
                e[(TP_RO + 2) * 64] = __float_as_int(o.z);
                e[(TP_RD + 0) * 64] = __float_as_int(d.x);
                e[(TP_RD + 1) * 64] = __float_as_int(d.y);
                e[(TP_RD + 2) * 64] = __float_as_int(d.z);
                e[(TP_RNG + 0) * 64] = (int)(uint32_t)r;
                e[(TP_RNG + 1) * 64] = (int)(uint32_t)(r >> 32);
                e[TP_TMIN * 64] = __float_as_int(it.t_min);
                e[TP_TMAX * 64] = __float_as_int(it.t_max);
                e[(TP_NT + 0) * 64] = __float_as_int(it.next_t[0]);
                e[(TP_NT + 1) * 64] = __float_as_int(it.next_t[1]);
                e[(TP_NT + 2) * 64] = __float_as_int(it.next_t[2]);
                e[(TP_DL + 0) * 64] = __float_as_int(it.delta_t[0]);
                e[(TP_DL + 1) * 64] = __float_as_int(it.delta_t[1]);
                e[(TP_DL + 2) * 64] = __float_as_int(it.delta_t[2]);
                e[TP_VOX * 64] = (it.mode & 0xff) == 2 ? (it.voxel[0] | (it.voxel[1] << 10) | (it.voxel[2] << 20)) : 0;
                e[TP_FL * 64] = (cur_tag << TP_TAG_SHIFT) | (dead ? TF_DEAD_NULL : 0) | ((it.mode & 0xff) == 2 ? TF_IT_LIVE : 0) | (it.mode & 0x700);
            }
            pool_n = take > 0 ? take : 0;
            cursor += pool_n;
            wave_lds_fence();
        }
        // ---- free lanes take a ready state from the pool ----
        if (free_m != 0ull && pool_n > 0) {
            const int rank = __popcll(free_m & lt_mask);
            if (state == TR_EMPTY && rank < pool_n) {
                const int* e = pool + (pool_n - 1 - rank);
                slot = (uint32_t)e[TP_SLOT * 64];
                ro = mk3(__int_as_float(e[(TP_RO + 0) * 64]), __int_as_float(e[(TP_RO + 1) * 64]), __int_as_float(e[(TP_RO + 2) * 64]));
                rd = mk3(__int_as_float(e[(TP_RD + 0) * 64]), __int_as_float(e[(TP_RD + 1) * 64]), __int_as_float(e[(TP_RD + 2) * 64]));
                rng = (uint64_t)(uint32_t)e[(TP_RNG + 0) * 64] | ((uint64_t)(uint32_t)e[(TP_RNG + 1) * 64] << 32);
                it_tmin = __int_as_float(e[TP_TMIN * 64]);
                it_tmax = __int_as_float(e[TP_TMAX * 64]);
                nt0 = __int_as_float(e[(TP_NT + 0) * 64]), nt1 = __int_as_float(e[(TP_NT + 1) * 64]), nt2 = __int_as_float(e[(TP_NT + 2) * 64]);
                dl0 = __int_as_float(e[(TP_DL + 0) * 64]), dl1 = __int_as_float(e[(TP_DL + 1) * 64]), dl2 = __int_as_float(e[(TP_DL + 2) * 64]);
                const int vox = e[TP_VOX * 64];
                vx = vox & 1023, vy = (vox >> 10) & 1023, vz = vox >> 20;
                fl = e[TP_FL * 64];
                rl_f = 1.0f;
                segi = 0;
                state = TR_BUSY;
            }
            const int free_n = __popcll(free_m);
            pool_n -= free_n < pool_n ? free_n : pool_n;
            wave_lds_fence();
        }
        if (__ballot(state == TR_BUSY) == 0ull) {
            if (pool_n == 0 && cursor >= n && !more) {
                if (res_n == 0) break;
                drain = true;
            }
            continue;
        }
        // ---- phase A: cheap steps (next majorant cell, free-flight sample) until the busy lanes hold a tentative collision or have
        //      run out of cells ----
#pragma unroll 1
        for (int adv = 0;; ++adv) {
            const bool need = state == TR_BUSY && (fl & (TF_PENDING | TF_SURVIVED)) == 0;
            if (__ballot(need) == 0ull) break;
            if (adv >= fr.delta_advance) {
                if (adv >= gate_cap) break;
                if (__popcll(__ballot(state == TR_BUSY && (fl & TF_PENDING) != 0)) >= gate_min) break;
            }
            HK_DBG(0, need);
            if (need) {
                if ((fl & TF_IN_SEG) == 0) {
                    HK_DBG(1, true);
                    // majorant_next (media.jl:625-729) of a DDA iterator, straight-line
                    if ((fl & TF_IT_LIVE) == 0 || segi >= 256 || it_tmin >= it_tmax)
                        fl |= TF_SURVIVED;   // ran out of segments with the path still alive
                    else {
                        const bool lxy = nt0 < nt1, lxz = nt0 < nt2, lyz = nt1 < nt2;
                        const bool ax0 = lxy & lxz, ax1 = (!lxy) & lyz;   // axis 0, axis 1, else axis 2
                        const float nt = ax0 ? nt0 : (ax1 ? nt1 : nt2);
                        const float stm = minf(nt, it_tmax);
                        const float rho = maj[vx + mrx * (vy + mry * vz)];
                        const float seg0 = it_tmin;
                        seg1 = stm;
                        const bool neg = (fl & (ax0 ? TF_NEG0 : (ax1 ? TF_NEG1 : TF_NEG2))) != 0;
                        const int v = (ax0 ? vx : (ax1 ? vy : vz)) + (neg ? -1 : 1);
                        const int lim = neg ? -1 : (ax0 ? mrx : (ax1 ? mry : mrz));
                        const float s = nt + (ax0 ? dl0 : (ax1 ? dl1 : dl2));
                        vx = ax0 ? v : vx;
                        vy = ax1 ? v : vy;
                        vz = (ax0 | ax1) ? vz : v;
                        nt0 = ax0 ? s : nt0;
                        nt1 = ax1 ? s : nt1;
                        nt2 = (ax0 | ax1) ? nt2 : s;
                        const bool out = v == lim;
                        fl = out ? (fl & ~TF_IT_LIVE) : fl;
                        it_tmin = out ? it_tmax : stm;
                        ++segi;
                        ++n_dda;
                        sm0 = sig_t * rho;
                        if (sm0 >= 1e-10f) {
                            t = seg0;
                            cur_o = ro + rd * t;
                            fl |= TF_IN_SEG;
                            k_in_seg = 0;
                        }
                    }
                }
                // a lane that has just entered a cell draws its first free flight in the same round
                if ((fl & (TF_IN_SEG | TF_SURVIVED)) == TF_IN_SEG) {
                    HK_DBG(2, true);
                    if (k_in_seg >= 1024)
                        fl &= ~TF_IN_SEG;
                    else {
                        ++k_in_seg;
                        const float u = lcg_next(rng);
                        pend_dt = -media_logf(maxf(1e-10f, 1.0f - u)) / sm0;
                        const float ts = t + pend_dt;
                        fl = ts >= seg1 ? (fl & ~TF_IN_SEG) : (fl | TF_PENDING);   // leaves the cell (T_maj / T_maj[1] = 1: nothing else changes) / tentative collision
                    }
                }
            }
        }
        // ---- phase B: the tentative collisions (medium lookup, absorb / scatter / null) ----
        HK_DBG(3, state == TR_BUSY && (fl & TF_PENDING) != 0);
        HK_DBG(4, state == TR_BUSY);
        if (state == TR_BUSY && (fl & TF_PENDING) != 0) {
            fl &= ~TF_PENDING;
            const float dt = pend_dt;
            const float Tm0 = media_expf((-dt) * sm0);
            const v3 p = cur_o + rd * dt;
            ++n_coll;
            const float d = sample_density<MM, BRICKS>(med, p);
            const float sa = a0 * d, ss = s0 * d;
            const float p_absorb = sa / sm0, p_scatter = ss / sm0;
            const float ue = lcg_next(rng);
            if (ue < p_absorb) {
                state = TR_EMPTY;  // absorbed
            } else if (ue < p_absorb + p_scatter) {
                if (last_depth)
                    state = TR_EMPTY;
                else {   // beta and r_u are rescaled by sigma_s T_maj / (sigma_s T_maj)[1] = 1: they stay as they are in the record
                    g.ray_o[slot] = make_float4(p.x, p.y, p.z, INF_F);  // the scattering vertex: k_scatter continues from here
                    state = TR_SCATTER;
                }
            } else {
                const float sn0 = maxf(sm0 - sa - ss, 0.0f);
                const float pdf = Tm0 * sn0;
                if (pdf > 1e-10f) {
                    rl_f = ((rl_f * Tm0) * sm0) * (1.0f / pdf);
                    t = t + dt;
                    cur_o = p;
                    if (fl & TF_DEAD_NULL) state = TR_EMPTY;
                } else
                    state = TR_EMPTY;
            }
        }
        if (state == TR_BUSY && (fl & TF_SURVIVED) != 0) state = ((fl & TF_DEAD_NULL) != 0 || last_depth) ? TR_EMPTY : TP_SURVIVED;
    }
#pragma unroll
    for (int tg = 0; tg < TP_TAGS; ++tg) close_seg(tg);
    stats += global_wave();
    HK_DBG_FLUSH(stats);
    wave_add(&stats->collisions, n_coll);
    wave_add(&stats->nvdb_collisions, (sc.media_mask >> HK_MEDIUM_NANOVDB) & 1 ? n_coll : 0u);   // (priced per scene: SURVEY 8d charges the NanoVDB tree walk)
    wave_add(&stats->dda_steps, n_dda);
}

// ---------------------------------------------------------------------------------------------------
// K5 + K6: direct lighting at a medium scattering vertex (light-BVH NEE with n = 0, HG evaluated with cos = wo.wi,
// medium-scatter.jl:15-138) and phase-function sampling (medium-scatter.jl:148-216).  First writer of this depth's shadow /
// next-ray segments.
// ---------------------------------------------------------------------------------------------------
template <bool FT>   // FT: this bounce's Sobol draws are table loads for every path (see k_shade)
__global__ void __launch_bounds__(256) k_scatter(DPathState st, DScene sc, DTables T, DFrame fr, DSobol sob, int depth, DStats* stats) {
    unsigned n_lnodes = 0, n_sv = 0;
    HK_FOR_EACH_WAVE_SEGMENT(gw, st, ticket_ptr(st, depth, TK_SCATTER), depth, Q_SCATTER) {
        const uint32_t* __restrict__ queue = st.scatter_q + (size_t)gw * st.wave_cap;
        const int n = *count_ptr(st, depth, Q_SCATTER, gw);
        const DPathGen g = st.gen[depth & 1], gn = st.gen[(depth + 1) & 1];
        const size_t seg = (size_t)gw * st.wave_cap;
        WavePos q_shadow{0}, q_next{0};   // first writer of this depth's shadow records and of the next generation
        for (int base = 0; base < n; base += 64) {
            int i = base + lane_id();
            bool active = i < n;
            uint32_t slot = active ? queue[i] : 0u;
            bool push_shadow = false, push_ray = false;
            // records of the two pushes (written after the wave-wide position is known)
            float4 shO = make_float4(0, 0, 0, 0), shD = shO, nD = shO, O = shO;
            S4 shLd = s4(0.0f), shRu = s4(0.0f), shRl = s4(0.0f), lambda = s4(0.0f), beta = s4(0.0f), r_u = s4(0.0f), n_rl = s4(0.0f);
            uint32_t pslot = 0, nflags = 0;
            if (active) {
                ++n_sv;
                O = g.ray_o[slot];
                const float4 D = g.ray_d[slot];
                v3 sp = mk3(O.x, O.y, O.z), wo = mk3(-D.x, -D.y, -D.z);
                const uint2 meta = g.meta[slot];
                pslot = meta.y;
                const int medium_idx = (int)(meta.x >> 16) - 1;
                const float sg = sc.media[medium_idx].g;
                lambda = ld_lambda(st, g, slot, pslot);
                beta = ld4(&g.beta[slot]);
                r_u = ld_ru(g, slot, false, st.compact != 0);
                int pix, k;
                split_slot(fr, pslot, pix, k);
                SobolCtx sctx = sobol_ctx_slot(sob, T.sobol, fr.x0, fr.y0, fr.tiles_x, pix, k, fr.first_sample + k * fr.sample_stride);
                const int base_dim = 6 + 7 * depth;
                if (sc.n_lights > 0) {
                    float light_select = sobol_1d<FT>(sctx, base_dim + 1);
                    float light_pmf;
                    int light_idx = bvh_sample_light(sc, sp, mk3(0, 0, 0), light_select, light_pmf, n_lnodes);
                    if (light_idx >= 1 && light_idx <= sc.n_lights && light_pmf > 0.0f) {
                        const DLight& sel = sc.lights[light_idx - 1];
                        v2 u_light = mk2(0.0f, 0.0f);
                        if (sel.kind >= HK_LIGHT_AMBIENT) u_light = sobol_2d<FT>(sctx, base_dim + 3);
                        LightSample ls = sample_light(sc, T, sel, sp, lambda, u_light);
                        if (ls.pdf > 0.0f && !is_black(ls.Li)) {
                            float phase_val = hg_p(sg, dot(wo, ls.wi));
                            if (phase_val > 0.0f) {
                                float light_pdf = ls.pdf * light_pmf;
                                float phase_pdf = ls.is_delta ? 0.0f : phase_val;
                                float tmx = ls.is_delta ? norm(ls.p_light - sp) - 0.001f : 1.0e6f;
                                shO = make_float4(sp.x, sp.y, sp.z, tmx);
                                shD = make_float4(ls.wi.x, ls.wi.y, ls.wi.z, __int_as_float(medium_idx));
                                shLd = beta * phase_val * ls.Li;
                                shRu = r_u * phase_pdf;
                                shRl = r_u * light_pdf;
                                push_shadow = true;
                            }
                        }
                    }
                }
                int new_depth = depth + 1;
                if (new_depth < fr.max_depth) {
                    v2 u = sobol_2d<FT>(sctx, base_dim + 6);
                    float ppdf;
                    v3 wi = sample_hg(sg, wo, u, ppdf);
                    if (ppdf > 0.0f) {
                        nD = make_float4(wi.x, wi.y, wi.z, D.w);  // ray_o already holds the vertex, t_max = Inf
                        n_rl = r_u / ppdf;
                        nflags = (uint32_t)new_depth | (1u << 9) | ((uint32_t)(medium_idx + 1) << 16);  // specular = false, any_non_specular = true
                        push_ray = true;
                    }
                }
            }
            const size_t ps = seg + (size_t)wp_push(q_shadow, push_shadow);
            if (push_shadow) {
                st.sh_o[ps] = shO;
                st.sh_d[ps] = shD;
                st4(&st.sh_Ld[ps], shLd);
                st_shadow_weights(st, ps, shRu, shRl);
                st.sh_slot[ps] = pslot;
            }
            const size_t pn = seg + (size_t)wp_push(q_next, push_ray);
            if (push_ray) {   // the continuing path's record, whole, at its position in the next generation
                gn.ray_o[pn] = O;
                if (st.compact) nD.w = n_rl.x;
                gn.ray_d[pn] = nD;
                st4(&gn.beta[pn], beta);
                st_ru_rl(gn, pn, r_u, n_rl, st.compact != 0);
                st_lambda(gn, pn, lambda);
                gn.meta[pn] = make_uint2(nflags, pslot);
            }
        }
        if (lane_id() == 0) {
            *count_ptr(st, depth, Q_SHADOW, gw) = q_shadow.count;
            *count_ptr(st, depth + 1, Q_RAY, gw) = q_next.count;
        }
    }
    stats += global_wave();
    wave_add(&stats->sc_light_nodes, n_lnodes);
    wave_add(&stats->scatter_vertices, n_sv);
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
// one escaped path: what it adds to its L (false: nothing)
__device__ __forceinline__ bool escaped_one(const DPathState& st, const DScene& sc, const DTables& T, const DPathGen& g, bool ones, uint32_t slot, S4& fin, uint32_t& pslot, int depth) {
    const uint2 meta = ld_meta(st, g, slot, depth);
    pslot = meta.y;
    S4 lambda = ld_lambda(st, g, slot, meta.y);
    S4 Le = s4(0.0f);
    v3 rd = mk3(0, 0, 1);
    bool have_dir = false;
    for (int li = 0; li < sc.n_lights; ++li) {
        const DLight& l = sc.lights[li];
        if (l.kind == HK_LIGHT_AMBIENT) Le = Le + l.scale * light_spectrum(l, lambda);
        if (l.kind == HK_LIGHT_ENVIRONMENT) {  // bilinear env(dir) * scale, illuminant uplift (lights.jl:408-419)
            if (!have_dir) {
                float4 D = g.ray_d[slot];
                rd = mk3(D.x, D.y, D.z);
                have_dir = true;
            }
            float4 t = env_eval(sc.envmaps[l.Le_tex], rd);
            Le = Le + eval_illuminant(coef_illuminant(T, t.x * l.Le_rgba[0], t.y * l.Le_rgba[1], t.z * l.Le_rgba[2]), lambda);
        }
    }
    S4 beta = ld_throughput(g.beta, slot, ones);
    S4 contribution = beta * Le;
    if (is_black(contribution)) return false;
    uint32_t fl = meta.x;
    int pdepth = (int)(fl & 0xff);
    bool specular = (fl >> 8) & 1u;
    S4 r_u = ld_ru(g, slot, ones, st.compact != 0);
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
        S4 rl = ld_rl(g, slot, ones, st.compact != 0) * choice * light_pdf;
        float den = average(r_u + rl);
        fin = den > 1e-10f ? contribution / den : contribution / average(r_u);
    }
    return true;
}
// UNROLL = 2 (round 6): a lane carries TWO queue entries through the chain of dependent loads (queue entry -> record -> environment texels ->
// spectrum table -> L) at once.  The kernel spends 0.81 of its wave time parked at s_waitcnt (profiles/r05_wavestate_sky.txt) with one
// entry per lane in flight; the two are different paths, so their additions to L do not meet and the film is the same bit for bit.
template <int UNROLL>
__device__ __forceinline__ void escaped_body(const DPathState& st, const DScene& sc, const DTables& T, int depth, int implicit_ones, const SegTickets& src) {
    HK_FOR_EACH_SEGMENT_FROM(gw, st, src) {
    const uint32_t* __restrict__ queue = st.escaped_q + (size_t)gw * st.wave_cap;
    const int n = *count_ptr(st, depth, Q_ESCAPED, gw);
    const DPathGen g = st.gen[depth & 1];
    const bool ones = depth == 0 && implicit_ones;
    int i = lane_id();
    if (UNROLL == 2) {
        for (; i + 64 < n; i += 128) {
            const uint32_t slot0 = queue[i], slot1 = queue[i + 64];
            S4 fin0, fin1;
            uint32_t p0, p1;
            const bool a0 = escaped_one(st, sc, T, g, ones, slot0, fin0, p0, depth);
            const bool a1 = escaped_one(st, sc, T, g, ones, slot1, fin1, p1, depth);
            S4 L0 = s4(0.0f), L1 = s4(0.0f);
            if (a0) L0 = ld4(&st.L[p0]);
            if (a1) L1 = ld4(&st.L[p1]);
            if (a0) st4(&st.L[p0], L0 + fin0);
            if (a1) st4(&st.L[p1], L1 + fin1);
        }
    }
    for (; i < n; i += 64) {
        S4 fin;
        uint32_t pslot;
        if (escaped_one(st, sc, T, g, ones, queue[i], fin, pslot, depth)) st4(&st.L[pslot], ld4(&st.L[pslot]) + fin);
    }
    }
}
template <int UNROLL>
__global__ void __launch_bounds__(256) k_escaped(DPathState st, DScene sc, DTables T, int depth, int implicit_ones) {
    escaped_body<UNROLL>(st, sc, T, depth, implicit_ones, seg_open(st, ticket_ptr(st, depth, TK_ESCAPED), st.dynamic_segments != 0, depth, Q_ESCAPED));
}

// ---------------------------------------------------------------------------------------------------
// K8 + K9 + K11 fused per material kind (surface-eval.jl:147-220, 250-341, 396-512).
// ---------------------------------------------------------------------------------------------------
// Register budget: 512 VGPRs per SIMD lane => 3 waves/SIMD need <= 168, 4 need <= 128.  The simple kinds sit a few registers
// above 168 without a hint; asking for 3 waves costs a handful of scratch spills and buys a third more latency hiding.
// The walk kinds (coated diffuse / transmission) are far above: they keep the default.
// ---------------------------------------------------------------------------------------------------
// Light selection of K9 as its own kernel, for scenes with a deep light BVH (DScene::num_bvh_lights >= HK_PRESELECT_MIN, no media).
// A descent of bvh_sample_light (bvh-light-sampler.jl:105-170) takes as many levels as the chosen light's leaf is deep — 12 to 25 in
// the 5*10^4-light scene — and inside k_shade every lane of a wave waits for the deepest: 49 % of the lanes active.  Here each
// lane is a little state machine (like the traversal kernels'): one level per round for the lanes that descend, finished lanes keep
// their result until enough of them wait (HK_SELECT_MIN_IDLE = 24: 8 / 16 / 24 / 32 idle lanes give 0.78 / 0.72 / 0.69 / 0.70 s of shading per
// 512-spp frame; computing the inputs in a dense pre-pass changed nothing), then write it and pull the next shading vertices of the segment — the
// entries of its per-kind queues, one kind after the other.  Same arithmetic per vertex as the fused form: position and shading normal
// from surface_at, the sample from dimension base + 1, node_importance in the same order; the result (light, pmf) goes to sel_light[entry].
// ---------------------------------------------------------------------------------------------------
#define HK_PRESELECT_MIN 64
#ifndef HK_SELECT_MIN_IDLE
#define HK_SELECT_MIN_IDLE 24
#endif
template <bool FT>
__global__ void __launch_bounds__(256) k_light_select(DPathState st, DScene sc, DTables T, DFrame fr, DSobol sob, int depth, uint32_t kinds_mask, int min_idle, DStats* stats) {
    const int lane = lane_id();
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    unsigned n_lnodes = 0;
    const DPathGen g = st.gen[depth & 1];
    const int base_dim = 6 + 7 * depth;
    const int ninf = sc.num_infinite_lights, nbvh = sc.num_bvh_lights;
    const bool has_bvh = nbvh > 0;
    const float p_inf = (float)ninf / (float)(ninf + (has_bvh ? 1 : 0));
    HK_FOR_EACH_WAVE_SEGMENT(gw, st, ticket_ptr(st, depth, TK_SELECT), depth, Q_RAY) {
        int kind = -1, n = 0, cursor = 0;
        const uint32_t* __restrict__ queue = st.mat_q;
        bool more = true;
        // per-lane descent state
        bool busy = false, done = false;
        uint32_t slot = 0, bits = 0, child = 0;
        int ni = 1, res_light = 0, lvl = 0;
        float ub = 0.0f, pmf = 0.0f, res_pmf = 0.0f;
        v3 p = mk3(0, 0, 0), nn = mk3(0, 0, 1);
        for (;;) {
            const unsigned long long run_m = __ballot(busy && !done);
            if (run_m == 0ull || (64 - __popcll(run_m) >= min_idle && (cursor < n || more))) {
                if (busy && done) {
                    st.sel_light[slot] = make_uint2((uint32_t)res_light, __float_as_uint(res_pmf));
                    busy = false;
                }
                while (more && cursor >= n) {   // the next kind present in the scene that has entries in this segment
                    ++kind;
                    while (kind < HK_MAX_KINDS && !(kinds_mask & (1u << kind))) ++kind;
                    if (kind >= HK_MAX_KINDS) {
                        more = false;
                        break;
                    }
                    queue = st.mat_q + ((size_t)kind * st.n_waves + gw) * st.wave_cap;
                    n = *count_ptr(st, depth, Q_MAT0 + kind, gw);
                    cursor = 0;
                }
                const unsigned long long want = __ballot(!busy);
                const int avail = n - cursor;
                const int rank = __popcll(want & lt_mask);
                if (!busy && rank < avail) {
                    slot = queue[cursor + rank];
                    const float4 H = st.hit[slot], O = ld_ray_o(st, g, slot, depth, (size_t)gw * st.wave_cap), D = g.ray_d[slot];
                    const Surface sf = surface_at(sc, __float_as_int(H.y), H.z, H.w, mk3(O.x, O.y, O.z), mk3(D.x, D.y, D.z), H.x);
                    p = sf.pi;
                    nn = sf.ns;
                    int pix, k;
                    split_slot(fr, ld_meta(st, g, slot, depth).y, pix, k);
                    const SobolCtx sctx = sobol_ctx_slot(sob, T.sobol, fr.x0, fr.y0, fr.tiles_x, pix, k, fr.first_sample + k * fr.sample_stride);
                    const float u = sobol_1d<FT>(sctx, base_dim + 1);
                    // bvh_sample_light's prologue: the infinite lights, then the root
                    busy = true;
                    done = true;
                    res_light = 0;
                    res_pmf = 0.0f;
                    if (ninf + nbvh > 0) {
                        if (ninf > 0 && u < p_inf) {
                            const float ur = u / p_inf;
                            int idx = (int)floorf(ur * (float)ninf);
                            idx = (idx < ninf - 1 ? idx : ninf - 1) + 1;
                            res_pmf = p_inf / (float)ninf;
                            res_light = sc.infinite_lights[idx - 1];
                        } else if (has_bvh) {
                            ub = ninf > 0 ? minf((u - p_inf) / (1.0f - p_inf), 0.99999994f) : minf(u, 0.99999994f);
                            pmf = 1.0f - p_inf;
                            ni = 1;
                            const DLightNode root = load_light_node(sc.lnodes, 0);
                            bits = root.bits, child = root.child1_or_light;
                            lvl = 0;
                            done = false;
                        }
                    }
                }
                const int want_n = __popcll(want);
                cursor += want_n < avail ? want_n : (avail > 0 ? avail : 0);
                if (__ballot(busy) == 0ull) {
                    if (cursor >= n && !more) break;
                    continue;
                }
            }
            // ---- one level of the descent for the lanes that are on their way ----
            if (busy && !done) {
                if (lvl >= 64) {   // the reference gives up after 64 levels (bvh-light-sampler.jl:126)
                    res_light = 0;
                    res_pmf = 0.0f;
                    done = true;
                } else if (bits & 2u) {
                    res_pmf = pmf;
                    res_light = (int)child;
                    done = true;
                } else {
                    ++lvl;
                    const DLightNode n0 = load_light_node(sc.lnodes, (int)child), n1 = load_light_node(sc.lnodes, (int)child + 1);
                    const float c0 = node_importance(n0, p, nn);
                    const float c1 = node_importance(n1, p, nn);
                    n_lnodes += 2;
                    if (c0 == 0.0f && c1 == 0.0f) {
                        res_light = 0;
                        res_pmf = 0.0f;
                        done = true;
                    } else {
                        const float p0 = c0 / (c0 + c1);
                        if (ub < p0) {
                            pmf *= p0;
                            ub = ub / p0;
                            bits = n0.bits, child = n0.child1_or_light;
                        } else {
                            pmf *= (1.0f - p0);
                            ub = (ub - p0) / (1.0f - p0);
                            bits = n1.bits, child = n1.child1_or_light;
                        }
                    }
                }
            }
        }
    }
    stats += global_wave();
    wave_add(&stats->light_nodes, n_lnodes);
}

// ---------------------------------------------------------------------------------------------------
// k_light_select re-shaped around a per-wave POOL in LDS (round 5; the round-4 re-shape of the cloud's tracking kernels carried over).
// The per-lane-refill kernel above ran its descent rounds 70 % full: finished lanes waited for 24 of their kind, and the per-vertex
// SET-UP (hit record, surface_at, the Sobol draw, the infinite-light prologue) ran for those ~25 lanes only.  Here
//   * the set-up runs for 64 queue entries AT ONCE whenever the pool is empty and a lane is free; vertices that need a descent leave
//     (entry, p, n, u_bvh) — 8 words — in the pool, the others (an infinite light chosen, no tree) are answered on the spot;
//   * a lane whose descent reaches its leaf writes sel_light[entry] and takes the next vertex from the pool IN THE SAME ROUND;
//   * results are indexed by entry, nothing is left behind in a segment: the wave streams segment after segment and kind after kind
//     (SegStream) and drains its lanes once, at the end of the launch.
// Arithmetic per vertex is that of bvh_sample_light, operation for operation (test_light_bvh_parity: pmf at 0 ulp;
// test_light_preselection_is_result_neutral: film bit-identical with the fused form and with the kernel above, HK_SELECT_POOL=0).
// ---------------------------------------------------------------------------------------------------
//   * desynchronised lanes each fetch their own 128-B pair every round — 64 different lines per load instruction, where the batches of
//     the kernel above still shared the top of the tree: the first pooled version issued 20 % fewer VALU instructions at 84 % lane
//     utilisation (67 %) and was SLOWER, waiting 51 % of its wave-cycles for memory (16 %).  So the top NTOP entries of the tree — the
//     first 9 levels in the pair order, half of an average descent — live in the block's LDS (one array per 16-byte part of a node, as
//     in NodeCache).
#ifndef HK_SELECT_NTOP
#define HK_SELECT_NTOP 512
#endif
#ifndef HK_SELECT_BLOCK
#define HK_SELECT_BLOCK 256
#endif
template <int NTOP>
HKD DLightNode lds_light_node(const lds_float4* top, int i) {
    const hk_f4v a = top[i], b = top[NTOP + i], c = top[2 * NTOP + i], d = top[3 * NTOP + i];
    DLightNode n;
    n.centre[0] = a.x, n.centre[1] = a.y, n.centre[2] = a.z, n.half_diag = a.w;
    n.r2 = b.x, n.w[0] = b.y, n.w[1] = b.z, n.w[2] = b.w;
    n.phi = c.x, n.cos_o = c.y, n.cos_e = c.z, n.sin_o = c.w;
    n.bits = __float_as_uint(d.x), n.child1_or_light = __float_as_uint(d.y);
    n.pad[0] = n.pad[1] = 0u;
    return n;
}
#ifndef HK_SELECT_WAVES
#define HK_SELECT_WAVES 4   // 40 KB of LDS per 4-wave block: four blocks per CU.  Measured (many-light frame, shade class, interleaved on one box):
#endif                      // per-lane refill 0.680 s; pool without the top cache 0.759; 256 / 512 / 4 waves 0.613; 512 / 512 / 6 waves (80 VGPRs, spills) 0.650;
                            // 512 / 512 / 4 waves 0.617; one 1024-thread block with the top 1024 entries 0.610
template <bool FT, int BLOCK = HK_SELECT_BLOCK, int NTOP = HK_SELECT_NTOP>
__global__ void __attribute__((amdgpu_flat_work_group_size(BLOCK, BLOCK), amdgpu_waves_per_eu(HK_SELECT_WAVES))) k_light_select_pool(DPathState st, DScene sc, DTables T, DFrame fr, DSobol sob, int depth, uint32_t kinds_mask, DStats* stats) {
    __shared__ uint32_t lds_pool[BLOCK / 64][8][64];   // [wave of the block][field][entry] (bit patterns): lane i reads entry head + rank(i) — consecutive words, no bank conflict
    __shared__ float4 lds_top[NTOP > 0 ? 4 * NTOP : 1];
    uint32_t (*pool)[64] = lds_pool[threadIdx.x >> 6];
    const lds_float4* top = (const lds_float4*)lds_top;
    const int n_top = NTOP > 0 ? (2 * sc.num_bvh_lights < NTOP ? 2 * sc.num_bvh_lights : NTOP) : 0;   // the tree of n lights has 2 n entries (entry 1 unused)
    if (NTOP > 0) {
        for (int i = threadIdx.x; i < n_top; i += BLOCK) {
            const float4* q = reinterpret_cast<const float4*>(sc.lnodes + i);
            lds_top[i] = q[0], lds_top[NTOP + i] = q[1], lds_top[2 * NTOP + i] = q[2], lds_top[3 * NTOP + i] = q[3];
        }
        __syncthreads();
    }
    const int lane = lane_id();
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    unsigned n_lnodes = 0;
    const DPathGen g = st.gen[depth & 1];
    const int base_dim = 6 + 7 * depth;
    const int ninf = sc.num_infinite_lights, nbvh = sc.num_bvh_lights;
    const bool has_bvh = nbvh > 0;
    const float p_inf = (float)ninf / (float)(ninf + (has_bvh ? 1 : 0));
    uint32_t root_bits = 2u, root_child = 0u;
    if (has_bvh) {
        const DLightNode root = load_light_node(sc.lnodes, 0);
        root_bits = __builtin_amdgcn_readfirstlane(root.bits), root_child = __builtin_amdgcn_readfirstlane(root.child1_or_light);
    }
    SegStream stream = stream_open(st, ticket_ptr(st, depth, TK_SELECT), false, depth, Q_RAY);
    int gw = -1, kind = HK_MAX_KINDS, n = 0, cursor = 0;   // the input: entries cursor .. n - 1 of `queue`, the per-kind queue of segment gw
    const uint32_t* __restrict__ queue = st.mat_q;
    bool more = true;
    int pool_n = 0, pool_head = 0;
    // per-lane descent
    bool busy = false;
    uint32_t slot = 0, bits = 0, child = 0;
    int lvl = 0;
    float ub = 0.0f, pmf = 0.0f;
    v3 p = mk3(0, 0, 0), nn = mk3(0, 0, 1);
    for (;;) {
        const unsigned long long free_m = __ballot(!busy);
        if (free_m != 0ull) {
            if (pool_head == pool_n && more) {
                // ---- the next (up to 64) entries of the input ----
                while (more && cursor >= n) {
                    ++kind;
                    while (kind < HK_MAX_KINDS && !(kinds_mask & (1u << kind))) ++kind;
                    if (kind >= HK_MAX_KINDS) {   // this segment's kinds are used up: the next segment
                        gw = stream_next(stream, st.n_waves);
                        if (gw >= st.n_waves) {
                            more = false;
                            break;
                        }
                        kind = -1;
                        continue;
                    }
                    queue = st.mat_q + ((size_t)kind * st.n_waves + gw) * st.wave_cap;
                    n = *count_ptr(st, depth, Q_MAT0 + kind, gw);
                    cursor = 0;
                }
                pool_n = pool_head = 0;
                if (more) {
                    const int take = n - cursor < 64 ? n - cursor : 64;
                    bool descend = false;
                    uint32_t e = 0;
                    float u_b = 0.0f;
                    v3 sp = mk3(0, 0, 0), sn = mk3(0, 0, 1);
                    if (lane < take) {
                        e = queue[cursor + lane];
                        const float4 H = st.hit[e], O = ld_ray_o(st, g, e, depth, (size_t)gw * st.wave_cap), D = g.ray_d[e];
                        const Surface sf = surface_at(sc, __float_as_int(H.y), H.z, H.w, mk3(O.x, O.y, O.z), mk3(D.x, D.y, D.z), H.x);
                        sp = sf.pi;
                        sn = sf.ns;
                        int pix, k;
                        split_slot(fr, ld_meta(st, g, e, depth).y, pix, k);
                        const SobolCtx sctx = sobol_ctx_slot(sob, T.sobol, fr.x0, fr.y0, fr.tiles_x, pix, k, fr.first_sample + k * fr.sample_stride);
                        const float u = sobol_1d<FT>(sctx, base_dim + 1);
                        // bvh_sample_light's prologue: the infinite lights, then the root
                        int res_light = 0;
                        float res_pmf = 0.0f;
                        if (ninf + nbvh > 0) {
                            if (ninf > 0 && u < p_inf) {
                                const float ur = u / p_inf;
                                int idx = (int)floorf(ur * (float)ninf);
                                idx = (idx < ninf - 1 ? idx : ninf - 1) + 1;
                                res_pmf = p_inf / (float)ninf;
                                res_light = sc.infinite_lights[idx - 1];
                            } else if (has_bvh) {
                                u_b = ninf > 0 ? minf((u - p_inf) / (1.0f - p_inf), 0.99999994f) : minf(u, 0.99999994f);
                                if (root_bits & 2u) {   // a tree of one light
                                    res_pmf = 1.0f - p_inf;
                                    res_light = (int)root_child;
                                } else
                                    descend = true;
                            }
                        }
                        if (!descend) st.sel_light[e] = make_uint2((uint32_t)res_light, __float_as_uint(res_pmf));
                    }
                    cursor += take;
                    const unsigned long long dm = __ballot(descend);
                    if (descend) {
                        const int at = __popcll(dm & lt_mask);
                        pool[0][at] = e;
                        pool[1][at] = __float_as_uint(sp.x), pool[2][at] = __float_as_uint(sp.y), pool[3][at] = __float_as_uint(sp.z);
                        pool[4][at] = __float_as_uint(sn.x), pool[5][at] = __float_as_uint(sn.y), pool[6][at] = __float_as_uint(sn.z);
                        pool[7][at] = __float_as_uint(u_b);
                    }
                    pool_n = __popcll(dm);
                    __builtin_amdgcn_wave_barrier();   // (the pool belongs to this wave alone: LDS operations of one wave complete in order)
                }
            }
            const int avail = pool_n - pool_head;
            if (avail > 0) {
                const int rank = __popcll(free_m & lt_mask);
                if (!busy && rank < avail) {
                    const int at = pool_head + rank;
                    slot = pool[0][at];
                    p = mk3(__uint_as_float(pool[1][at]), __uint_as_float(pool[2][at]), __uint_as_float(pool[3][at]));
                    nn = mk3(__uint_as_float(pool[4][at]), __uint_as_float(pool[5][at]), __uint_as_float(pool[6][at]));
                    ub = __uint_as_float(pool[7][at]);
                    pmf = 1.0f - p_inf;
                    bits = root_bits, child = root_child;
                    lvl = 0;
                    busy = true;
                }
                const int free_n = __popcll(free_m);
                pool_head += free_n < avail ? free_n : avail;
            }
            if (__ballot(busy) == 0ull) {
                if (!more && pool_head == pool_n) break;
                continue;
            }
        }
        // ---- one level of the descent for every lane that is on its way (a lane that took a vertex above starts at once) ----
        if (busy) {
            int res_light = -1;
            float res_pmf = 0.0f;
            if (lvl >= 64) {   // the reference gives up after 64 levels (bvh-light-sampler.jl:126)
                res_light = 0;
            } else {
                ++lvl;
                DLightNode n0, n1;
                if (NTOP > 0 && (int)child + 1 < n_top) {
                    n0 = lds_light_node<NTOP>(top, (int)child), n1 = lds_light_node<NTOP>(top, (int)child + 1);
                    asm volatile("" : "+v"(n0.bits));   // (keeps the two arms apart: a select of generic pointers would become one flat load)
                } else
                    n0 = load_light_node(sc.lnodes, (int)child), n1 = load_light_node(sc.lnodes, (int)child + 1);
                const float c0 = node_importance(n0, p, nn);
                const float c1 = node_importance(n1, p, nn);
                n_lnodes += 2;
                if (c0 == 0.0f && c1 == 0.0f) {
                    res_light = 0;
                } else {
                    const float p0 = c0 / (c0 + c1);
                    if (ub < p0) {
                        pmf *= p0;
                        ub = ub / p0;
                        bits = n0.bits, child = n0.child1_or_light;
                    } else {
                        pmf *= (1.0f - p0);
                        ub = (ub - p0) / (1.0f - p0);
                        bits = n1.bits, child = n1.child1_or_light;
                    }
                    if ((bits & 2u) && lvl < 64) {   // the chosen child is a leaf: its light, with the pmf of the way down (a leaf 64 levels down is never looked at: bvh-light-sampler.jl:126)
                        res_light = (int)child;
                        res_pmf = pmf;
                    }
                }
            }
            if (res_light >= 0) {
                st.sel_light[slot] = make_uint2((uint32_t)res_light, __float_as_uint(res_pmf));
                busy = false;
            }
        }
    }
    stats += global_wave();
    wave_add(&stats->light_nodes, n_lnodes);
}

#ifndef HK_SHADE_WAVES_MATTE
#define HK_SHADE_WAVES_MATTE 4
#endif
#ifndef HK_SHADE_WAVES
#define HK_SHADE_WAVES 3
#endif
// K8 for one flagged vertex (surface-eval.jl:147-220): L += beta * Le / MIS denominator.
template <bool TWO_PLANES, bool SIMPLE>
HKD void shade_emission(const DPathState& st, const DPathGen& g, bool ones, const DScene& sc, const DTables& T, uint32_t slot, unsigned& n_lnodes, int depth, size_t seg) {
    const float4 H = st.hit[slot];
    const float4 O = ld_ray_o(st, g, slot, depth, seg), D = g.ray_d[slot];
    const v3 ro = mk3(O.x, O.y, O.z), rd = mk3(D.x, D.y, D.z);
    const float t_hit = H.x;
    const int prim = __float_as_int(H.y);
    const DTriMeta meta = tri_meta(sc, prim);
    if (meta.arealight <= 0) return;
    const Surface sf = surface_at(sc, prim, H.z, H.w, ro, rd, t_hit);
    const v3 wo = -rd;
    const uint2 pmeta = ld_meta(st, g, slot, depth);
    const S4 lambda = ld_lambda(st, g, slot, pmeta.y);
    const DLight& light = sc.lights[meta.arealight - 1];
    S4 Le = arealight_Le<TWO_PLANES, SIMPLE>(sc, T, light, wo, sf.n, sf.uv, lambda);
    if (is_black(Le)) return;
    const S4 beta = ld_throughput(g.beta, slot, ones), r_u = ld_ru(g, slot, ones, st.compact != 0), r_l = ld_rl(g, slot, ones, st.compact != 0);
    const uint32_t fl = pmeta.x;
    const int pdepth = (int)(fl & 0xff);
    const bool specular_bounce = (fl >> 8) & 1u;
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
    st4(&st.L[pmeta.y], ld4(&st.L[pmeta.y]) + fin);
}

template <int KIND>
struct ShadeWaves {
    // Matte runs faster at 4 waves per SIMD with 224 B of scratch than at 3 with 48 B (Cornell k_shade -5 %); the kinds with larger
    // BSDFs lose more to the spills than the fourth wave returns (sky: conductor + glass +7 %)
    static constexpr int value = (KIND == HK_MAT_COATED_DIFFUSE || KIND == HK_MAT_COATED_DIFFUSE_TRANSMISSION) ? 1 : (KIND == HK_MAT_MATTE ? HK_SHADE_WAVES_MATTE : HK_SHADE_WAVES);
};
#ifndef HK_SHADE_MIN_WAVES
#define HK_SHADE_MIN_WAVES 1
#endif
// SIMPLE (instantiated for Matte): the scene has no ambient / environment light and no texture of any kind (DScene::simple_lights)
// FT ("full tables"; Matte, Mirror, Glass, Conductor): every Sobol draw of this bounce is in the sampler's two tables for every path of the launch (the host
// checks DSobol::lo_rows against the rows of this depth): the draws are two loads each and the digit-hashing fallback — three 64-bit
// hash loops inlined at each of the five draw sites — is not in the kernel at all.
// PRE: the light of this vertex's next-event estimation was chosen by k_light_select (scenes with a deep light BVH): the descent —
// three quarters of this kernel's time in the 5*10^4-light scene, run at 49 % lane utilisation because leaf depths differ — is not here.
template <int KIND, bool SIMPLE, bool FT, bool PRE>
__device__ __forceinline__ void shade_body(const DPathState& st, const DScene& sc, const DTables& T, const DFrame& fr, const DSobol& sob, int depth, int first_kind, DStats* stats, const SegTickets& src,
                                           uint32_t* __restrict__ elist) {   // elist: this wave's 128 LDS words (slots of flagged — emissive-hit — vertices waiting for the dense K8 pass)
    const int lane = lane_id();
    unsigned n_vertices = 0, n_lnodes = 0;
    HK_FOR_EACH_SEGMENT_FROM(gw, st, src) {
    const uint32_t* __restrict__ queue = st.mat_q + ((size_t)KIND * st.n_waves + gw) * st.wave_cap;
    const int n = *count_ptr(st, depth, Q_MAT0 + KIND, gw);
    const DPathGen g = st.gen[depth & 1], gn = st.gen[(depth + 1) & 1];
    const size_t seg = (size_t)gw * st.wave_cap;
    const bool ones = depth == 0 && fr.implicit_ones;
    // several kinds append to the same shadow-record / next-generation segments: continue from the counts left by the kinds before
    WavePos q_shadow{0}, q_next{0};
    if (!first_kind) {
        q_shadow.count = *count_ptr(st, depth, Q_SHADOW, gw);
        q_next.count = *count_ptr(st, depth + 1, Q_RAY, gw);
    }
    // ---- K8 first, densely: emission from area-light hits, MIS against the light-BVH pmf (surface-eval.jl:147-220).  The trace
    //      kernels flag such hits in mat_id; their entries are gathered in a wave-private LDS list and handled 64 at a time.  Done
    //      inline in the main pass, a wave walks the light BVH (bvh_pmf, ~2 log2 n node evaluations) whenever ANY of its 64
    //      vertices sits on an emitter: with 5 % emissive faces that is 96 % of the waves at 5 % lane utilisation. ----
    {
        int n_emit = 0;
        for (int base = 0; base < n || n_emit > 0; base += 64) {
            if (base < n) {
                const int i = base + lane;
                const uint32_t cand = i < n ? queue[i] : 0u;
                const bool em = i < n && (st.mat_id[cand] & HK_MAT_EMISSIVE_BIT) != 0;
                const unsigned long long m = __ballot(em);
                if (em) elist[n_emit + __popcll(m & ((1ull << lane) - 1ull))] = cand;
                n_emit += __popcll(m);
                if (n_emit < 64 && base + 64 < n) continue;
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            const int take = n_emit < 64 ? n_emit : 64;
            if (lane < take) shade_emission<KIND == HK_MAT_MATTE, SIMPLE>(st, g, ones, sc, T, elist[lane], n_lnodes, depth, seg);
            const int rest = n_emit - take;
            const uint32_t moved = lane < rest ? elist[64 + lane] : 0u;
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            if (lane < rest) elist[lane] = moved;
            n_emit = rest;
        }
    }
    for (int base = 0; base < n; base += 64) {
        int i = base + lane;
        bool active = i < n;
        uint32_t slot = active ? queue[i] : 0u;   // generation index of the path (the kind queue is a sorted subset of the segment)
        bool push_shadow = false, push_ray = false;
        // state shared by the two halves of the vertex (NEE, then BSDF sampling): loaded once
        float4 H = make_float4(0, 0, 0, 0);
        Surface sf;
        v3 wo = mk3(0, 0, 1);
        DTriMeta meta{0u, 0u, 0u};
        S4 lambda = s4(0.0f), beta = s4(0.0f), r_u = s4(0.0f), kd_matte = s4(0.0f);
        uint32_t fl = 0, pslot = 0;
        int pdepth = 0, medium = -1;
        bool any_non_specular = false;
        SobolCtx sctx;
        // every path in the depth-d queues has work.depth == d, so the dimension (and its scramble hashes) is
        // wave-uniform: derived from the kernel argument it stays in scalar registers
        const int base_dim = 6 + 7 * depth;
        float4 shO = make_float4(0, 0, 0, 0), shD = shO;
        S4 shLd = s4(0.0f), shRu = s4(0.0f), shRl = s4(0.0f);
        if (active) {
            ++n_vertices;
            H = stream_ld(&st.hit[slot]);
            float4 O = ld_ray_o(st, g, slot, depth, seg), D = stream_ld(&g.ray_d[slot]);
            v3 ro = mk3(O.x, O.y, O.z), rd = mk3(D.x, D.y, D.z);
            float t_hit = H.x;
            int prim = __float_as_int(H.y);
            sf = surface_at(sc, prim, H.z, H.w, ro, rd, t_hit);
            wo = -rd;
            meta = tri_meta(sc, prim);
            const uint2 pmeta = ld_meta(st, g, slot, depth);
            lambda = ld_lambda(st, g, slot, pmeta.y);
            if (ones)
                beta = s4(1.0f);
            else {
                const float4 bv = stream_ld(&g.beta[slot]);
                beta = s4(bv.x, bv.y, bv.z, bv.w);
            }
            r_u = ld_ru(g, slot, ones, st.compact != 0);
            fl = pmeta.x;
            pslot = pmeta.y;
            pdepth = (int)(fl & 0xff);
            any_non_specular = (fl >> 9) & 1u;
            medium = (int)(fl >> 16) - 1;

            // pixel coordinates for the Sobol dimensions of this bounce (volpath.jl:252-262, Q19)
            int pix, k;
            split_slot(fr, pslot, pix, k);
            sctx = sobol_ctx_slot(sob, T.sobol, fr.x0, fr.y0, fr.tiles_x, pix, k, fr.first_sample + k * fr.sample_stride);

            if (KIND == HK_MAT_MATTE) kd_matte = matte_kd<SIMPLE>(sc, T, sc.materials[st.mat_id[slot] & ~HK_MAT_EMISSIVE_BIT], TexCtx(sf.uv, meta.prim_index, H.z, H.w), lambda);
#ifdef HK_ABLATE
            // cost attribution by duplication (tools/ablate.md): stage HK_ABLATE runs a second time on laundered inputs, its result is
            // kept alive but unused; the k_shade time delta against the plain build is that stage's cost.  Never defined in the product.
            {
                float sink = 0.0f;
                if (HK_ABLATE == 1) {   // every Sobol draw of a matte vertex
                    uint32_t ps2 = pslot;
                    int d2 = base_dim;
                    asm volatile("" : "+v"(ps2));
                    asm volatile("" : "+s"(d2));
                    int pix2, k2;
                    split_slot(fr, ps2, pix2, k2);
                    SobolCtx c2 = sobol_ctx_slot(sob, T.sobol, fr.x0, fr.y0, fr.tiles_x, pix2, k2, fr.first_sample + k2 * fr.sample_stride);
                    v2 a = sobol_2d(c2, d2 + 3), b = sobol_2d(c2, d2 + 6);
                    sink = sobol_1d(c2, d2 + 1) + a.x + a.y + b.x + b.y + sobol_1d(c2, d2 + 7);
                }
                if (HK_ABLATE == 2 && sc.n_lights > 0) {   // light-BVH descent
                    float u2 = H.z, pmf2;
                    asm volatile("" : "+v"(u2));
                    unsigned dummy = 0;
                    sink = (float)bvh_sample_light(sc, sf.pi, sf.ns, u2, pmf2, dummy) + pmf2;
                }
                if (HK_ABLATE == 3 && sc.n_lights > 0) {   // sample_light on a pseudo-random light
                    uint32_t li = pslot;
                    asm volatile("" : "+v"(li));
                    LightSample l2 = sample_light(sc, T, sc.lights[li % (uint32_t)sc.n_lights], sf.pi, lambda, mk2(H.z, H.w));
                    sink = l2.pdf + l2.Li.x + l2.Li.w + l2.wi.x + l2.p_light.z;
                }
                if (HK_ABLATE == 4) {   // matte_kd (spectral reflectance at four wavelengths)
                    S4 l2 = lambda;
                    asm volatile("" : "+v"(l2.x), "+v"(l2.y), "+v"(l2.z), "+v"(l2.w));
                    S4 k2 = matte_kd(sc, T, sc.materials[st.mat_id[slot] & ~HK_MAT_EMISSIVE_BIT], TexCtx(sf.uv, meta.prim_index, H.z, H.w), l2);
                    sink = k2.x + k2.y + k2.z + k2.w;
                }
                if (HK_ABLATE == 5) {   // surface_at
                    float b2 = H.z;
                    asm volatile("" : "+v"(b2));
                    Surface s2 = surface_at(sc, prim, b2, H.w, ro, rd, t_hit);
                    sink = s2.pi.x + s2.n.y + s2.ns.z + s2.uv.x + s2.area;
                }
                if (HK_ABLATE == 6) {   // eval_matte_kd + sample_matte_kd on laundered directions
                    v3 w2 = wo;
                    asm volatile("" : "+v"(w2.x), "+v"(w2.y), "+v"(w2.z));
                    float p2;
                    S4 f2 = eval_matte_kd(kd_matte, w2, sf.n, sf.ns, p2);
                    BSDFSample s2 = sample_matte_kd(sc, sc.materials[st.mat_id[slot] & ~HK_MAT_EMISSIVE_BIT], kd_matte, w2, sf.ns, TexCtx(sf.uv, meta.prim_index, H.z, H.w), mk2(H.z, H.w));
                    sink = f2.x + f2.w + p2 + s2.pdf + s2.wi.x + s2.f.y;
                }
                asm volatile("" : : "v"(sink));
            }
#endif

            // ---- K9: next-event estimation through the light BVH ----
            if (sc.n_lights > 0) {
                const DMaterial& mat = sc.materials[st.mat_id[slot] & ~HK_MAT_EMISSIVE_BIT];
                float light_pmf;
                int light_idx;
                if constexpr (PRE) {
                    const uint2 pre = st.sel_light[slot];
                    light_idx = (int)pre.x;
                    light_pmf = __uint_as_float(pre.y);
                } else {
                    float light_select = sobol_1d<FT>(sctx, base_dim + 1);
                    light_idx = bvh_sample_light(sc, sf.pi, sf.ns, light_select, light_pmf, n_lnodes);
                }
                if (light_idx >= 1 && light_idx <= sc.n_lights && light_pmf > 0.0f) {
                    const DLight& sel = sc.lights[light_idx - 1];
                    // delta lights ignore the 2-D sample (lights.jl:39-131): draw it only for lights that use it
                    v2 u_light = mk2(0.0f, 0.0f);
                    if (sel.kind >= HK_LIGHT_AMBIENT) u_light = sobol_2d<FT>(sctx, base_dim + 3);
                    LightSample ls = sample_light<KIND == HK_MAT_MATTE, SIMPLE>(sc, T, sel, sf.pi, lambda, u_light);
                    if (ls.pdf > 0.0f && !is_black(ls.Li)) {
                        float bsdf_pdf;
                        S4 f = KIND == HK_MAT_MATTE ? eval_matte_kd(kd_matte, wo, ls.wi, sf.ns, bsdf_pdf)
                                                    : eval_bsdf<KIND>(sc, T, mat, wo, ls.wi, sf.ns, TexCtx(sf.uv, meta.prim_index, H.z, H.w), lambda, bsdf_pdf);
                        if (!is_black(f)) {
                            float ct = fabsf(dot(ls.wi, sf.ns));
                            S4 Ld = beta * f * ls.Li * ct;
                            if (!is_black(Ld)) {
                                v3 off = 1e-4f * sf.ns;  // NB: shading normal (Q9)
                                v3 so = dot(ls.wi, sf.ns) > 0.0f ? sf.pi + off : sf.pi - off;
                                v3 tl = ls.p_light - so;
                                float tmax = sqrtf(dot(tl, tl)) - 1e-3f;
                                float nbp = ls.is_delta ? 0.0f : bsdf_pdf;
                                shO = make_float4(so.x, so.y, so.z, tmax);
                                shD = make_float4(ls.wi.x, ls.wi.y, ls.wi.z, __int_as_float(medium));
                                shLd = Ld;
                                shRu = r_u * nbp;
                                shRl = (r_u * ls.pdf) * light_pmf;
                                push_shadow = true;
                            }
                        }
                    }
                }
            }
        }
        // the shadow record goes to its position in the segment's shadow queue: k_shadow streams the records
        {
            const size_t ps = seg + (size_t)wp_push(q_shadow, push_shadow);
            if (push_shadow && st.sh_final) {   // slim record (shadow_contribute_final): the denominator beside the direction instead of two weights
                const S4 mis = s4(shRu.x) * s4(1.0f) + s4(shRl.x) * s4(1.0f);
                shD.w = average(mis);
                stream_st(&st.sh_o[ps], shO);
                stream_st(&st.sh_d[ps], shD);
                stream_st(&st.sh_Ld[ps], shLd);
                stream_st(&st.sh_slot[ps], pslot);
            } else if (push_shadow) {
                stream_st(&st.sh_o[ps], shO);
                stream_st(&st.sh_d[ps], shD);
                stream_st(&st.sh_Ld[ps], shLd);
                if (st.compact) {
#if HK_NT
                    __builtin_nontemporal_store((hk_f2){shRu.x, shRl.x}, (hk_f2*)(reinterpret_cast<float2*>(st.sh_ru) + ps));
#else
                    reinterpret_cast<float2*>(st.sh_ru)[ps] = make_float2(shRu.x, shRl.x);
#endif
                } else
                    st_shadow_weights(st, ps, shRu, shRl);
                stream_st(&st.sh_slot[ps], pslot);
            }
        }
        // ---- K11: BSDF sampling, throughput, Russian roulette, continuation ray ----
        float4 nO = make_float4(0, 0, 0, 0), nD = nO;
        S4 nb = s4(0.0f), nrl = s4(0.0f);
        uint32_t nflags = 0;
        if (active) {
            int new_depth = pdepth + 1;
            if (new_depth < fr.max_depth) {
                const DMaterial& mat = sc.materials[st.mat_id[slot] & ~HK_MAT_EMISSIVE_BIT];
                const DMediumInterface mi = sc.mis[meta.mi];
                // the 1-D component sample is read only by BSDFs that choose a lobe (Glass and the layered kinds)
                float uc = (KIND == HK_MAT_GLASS || KIND > HK_MAT_CONDUCTOR) && KIND != HK_MAT_FALLBACK ? sobol_1d<FT>(sctx, base_dim + 4) : 0.0f;
                v2 u = (KIND == HK_MAT_MIRROR || KIND == HK_MAT_GLASS || KIND == HK_MAT_THIN_DIELECTRIC) ? mk2(0.0f, 0.0f) : sobol_2d<FT>(sctx, base_dim + 6);
                bool regularize = fr.regularize && any_non_specular;
                BSDFSample s = KIND == HK_MAT_MATTE ? sample_matte_kd<SIMPLE>(sc, mat, kd_matte, wo, sf.ns, TexCtx(sf.uv, meta.prim_index, H.z, H.w), u)
                                                    : sample_bsdf<KIND>(sc, T, mat, wo, sf.ns, TexCtx(sf.uv, meta.prim_index, H.z, H.w), lambda, u, uc, regularize);
                if (s.pdf > 0.0f && !is_black(s.f)) {
                    float ct = fabsf(dot(s.wi, sf.ns));
                    nb = s.is_specular ? beta * s.f : beta * s.f * ct / s.pdf;
                    nrl = s.is_specular ? r_u : r_u / s.pdf;
                    bool cont = true;
                    if (new_depth > 3) {  // russian_roulette_spectral, min_depth fixed at 3 (Q7)
                        float rr = sobol_1d<FT>(sctx, base_dim + 7);
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
                        nO = make_float4(no.x, no.y, no.z, INF_F);
                        nD = make_float4(s.wi.x, s.wi.y, s.wi.z, 0.0f);  // time = 0 (Q17)
                        bool ans = any_non_specular || !s.is_specular;
                        nflags = (uint32_t)new_depth | ((s.is_specular ? 1u : 0u) << 8) | ((ans ? 1u : 0u) << 9) | ((uint32_t)(new_medium + 1) << 16);
                        push_ray = true;
                    }
                }
            }
        }
        {   // the continuing path's record, whole, at its position in the next generation (r_u is unchanged by a surface event)
            const size_t pn = seg + (size_t)wp_push(q_next, push_ray);
            if (push_ray) {
                stream_st(&gn.ray_o[pn], nO);
                if (st.compact) nD.w = nrl.x;   // (compact: r_l travels beside the direction)
                stream_st(&gn.ray_d[pn], nD);
                stream_st(&gn.beta[pn], nb);
                st_ru_rl(gn, pn, r_u, nrl, st.compact != 0);
                st_lambda(gn, pn, lambda);
                st_meta(st, gn, pn, nflags, pslot);
            }
        }
    }
    if (lane == 0) {
        *count_ptr(st, depth, Q_SHADOW, gw) = q_shadow.count;
        *count_ptr(st, depth + 1, Q_RAY, gw) = q_next.count;
    }
    }
    stats += global_wave();
    wave_add(&stats->vertices, n_vertices);
    wave_add(&stats->light_nodes, n_lnodes);
}
template <int KIND, bool SIMPLE = false, bool FT = false, bool PRE = false>
__global__ void __attribute__((amdgpu_flat_work_group_size(256, 256), amdgpu_waves_per_eu(ShadeWaves<KIND>::value))) k_shade(DPathState st, DScene sc, DTables T, DFrame fr, DSobol sob, int depth, int first_kind, DStats* stats) {
    __shared__ uint32_t emit_list[4 * 128];
    shade_body<KIND, SIMPLE, FT, PRE>(st, sc, T, fr, sob, depth, first_kind, stats, seg_open(st, ticket_ptr(st, depth, TK_SHADE0 + KIND), st.dynamic_segments != 0, depth, Q_MAT0 + KIND),
                                      emit_list + (threadIdx.x >> 6) * 128);
}

// ---------------------------------------------------------------------------------------------------
// K10: shadow rays (intersection.jl:302-406, 565-600).
//   k_shadow       scenes without media whose surfaces are all opaque: one any-hit cast per ray.
//   k_shadow_walk  the general walk through medium-transition / alpha surfaces (<= 10 segments) with ratio tracking
//                  (intersection.jl:422-542) in the medium segments.  Like k_track it is a per-lane state machine with
//                  wave-private refill: a lane alternates between "needs a cast" and "tracking steps", casts are done
//                  together for all lanes that need one, and finished lanes pull the next shadow ray of the wave's queue.
// ---------------------------------------------------------------------------------------------------
// rec: index of the shadow record (segment * wave_cap + position in the segment's shadow queue)
template <class W> HKD W wone();
template <> HKD float wone<float>() { return 1.0f; }
template <> HKD S4 wone<S4>() { return s4(1.0f); }
HKD S4 wide(S4 v) { return v; }
HKD S4 wide(float v) { return s4(v); }
HKD bool is_black(float v) { return v == 0.0f; }
template <bool COMPACT>
HKD void shadow_contribute(const DPathState& st, uint32_t rec, S4 T_ray, S4 tr_u, S4 tr_l) {
    if (is_black(T_ray)) return;
    S4 w_u, w_l;
    ld_shadow_weights<COMPACT>(st, rec, w_u, w_l);
    S4 mis = w_u * tr_u + w_l * tr_l;
    float den = average(mis);
    if (den > 1e-10f) {
        S4 fin = ld4(&st.sh_Ld[rec]) * T_ray / den;
        if (!is_black(fin)) {
            // (this read-modify-write of one path's 16 B costs k_shadow a quarter of its time in the Cornell box — 15.4 -> 11.4 ms per
            // frame without it; float atomics: 47 ms; non-temporal loads of the records around it, or the path slot read with the ray and
            // weights / Ld / L loaded together (one memory round trip instead of four): +-0 — it is the traffic, not the latency)
            const uint32_t pslot = st.sh_slot[rec];
            st4(&st.L[pslot], ld4(&st.L[pslot]) + fin);
        }
    }
}

// SLIM SHADOW RECORDS (round 6; DPathState::sh_final, opaque scenes without media).  Without media a shadow ray's transmittance and its two
// track weights are 1, so the denominator of what an unoccluded ray adds — average(w_u * 1 + w_l * 1) — is known when the record is
// written: k_shade forms it (shadow_contribute's operations in its order: the same bits) and stores it in the unused fourth word of sh_d;
// the two weights (8 B) are neither written nor read — a 56-byte record instead of 60 and one stream less in k_shadow, which still does
// the division.  (First version: the DIVISION in k_shade too and the path slot beside the direction, 48 B — k_shadow -9 %, but the four
// correctly rounded divisions cost k_shade<Matte> +0.8 ms per Cornell frame, as much as the bytes returned: LAB_NOTEBOOK.md round 6.)
HKD void shadow_contribute_final(const DPathState& st, uint32_t rec, float den) {
    if (den > 1e-10f) {
        const S4 fin = ld4(&st.sh_Ld[rec]) * s4(1.0f) / den;
        if (!is_black(fin)) {
            const uint32_t pslot = st.sh_slot[rec];
            st4(&st.L[pslot], ld4(&st.L[pslot]) + fin);
        }
    }
}
// the layout of the shadow weights is a run-time fact where grey media may or may not use the compact records (k_walk_pool)
HKD void shadow_contribute_rt(const DPathState& st, uint32_t rec, S4 T_ray, S4 tr_u, S4 tr_l) {
    if (st.compact)
        shadow_contribute<true>(st, rec, T_ray, tr_u, tr_l);
    else
        shadow_contribute<false>(st, rec, T_ray, tr_u, tr_l);
}

template <bool COUNT, int NC, bool QN = false>
__device__ __forceinline__ void shadow_body(const DPathState& st, const DScene& sc, int depth, DStats* stats, SegStream stream, int* __restrict__ stack, const NodeCache& cache) {
    const int lane = lane_id();
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    const int DONE = (int)0x80000000;
    unsigned n_nodes = 0, n_tris = 0, n_casts = 0, n_hits = 0;
    HK_DBG_DECL
    uint32_t seg = 0;   // current segment's shadow records: entries seg .. seg + n - 1, streamed in order
    int n = 0, cursor = 0;
    bool more = true;
    bool have = false;
    uint32_t slot = 0;
    float rec_den = 0.0f;
    LaneRay r;
    r.cur = r.pend = DONE;
    for (;;) {
        const unsigned long long run_m = __ballot(have && r.cur != DONE);
        if (run_m == 0ull || (64 - __popcll(run_m) >= HK_TRACE_MIN_IDLE && (cursor < n || more))) {
            if (have && r.cur == DONE) {   // finished: an unoccluded shadow ray delivers its contribution
                if (r.best.prim < 0) {
                    if (st.sh_final)
                        shadow_contribute_final(st, slot, rec_den);
                    else
                        shadow_contribute<true>(st, slot, s4(1.0f), s4(1.0f), s4(1.0f));
                } else
                    ++n_hits;
                have = false;
            }
            while (more && cursor >= n) {   // this segment is used up: go on with the next one, the lanes in flight keep running
                const int gw = stream_next(stream, st.n_waves);
                if (gw >= st.n_waves) {
                    more = false;
                    break;
                }
                seg = (uint32_t)gw * (uint32_t)st.wave_cap;
                n = *count_ptr(st, depth, Q_SHADOW, gw);
                cursor = 0;
            }
            const unsigned long long want = __ballot(!have);
            const int avail = n - cursor;
            const int rank = __popcll(want & lt_mask);
            if (!have && rank < avail) {
                slot = seg + (uint32_t)(cursor + rank);
                float4 O = stream_ld(&st.sh_o[slot]), D = stream_ld(&st.sh_d[slot]);
                rec_den = D.w;   // (slim records: the denominator travels beside the direction)
                if (O.w >= 1e-6f) {   // a degenerate shadow ray is simply not visible
                    ++n_casts;
                    lane_ray_start<QN>(r, sc, mk3(O.x, O.y, O.z), mk3(D.x, D.y, D.z), O.w);
                    have = true;
                }
            }
            const int want_n = __popcll(want);
            cursor += want_n < avail ? want_n : (avail > 0 ? avail : 0);
            if (__ballot(have) == 0ull) {
                if (cursor >= n && !more) break;
                continue;
            }
        }
#ifdef HK_DEBUG_UTIL
        HK_DBG(5, have && r.cur != DONE);      // rounds: lanes with a shadow ray in flight (probes 3 / 4: its node steps / leaf phases)
        lane_ray_round<true, COUNT, NC, HK_POSTPONE_ANYHIT != 0, false, QN>(r, have && r.cur != DONE, sc, stack, lane, n_nodes, n_tris, cache, cursor < n || more, dbg_ + 6);
#else
        lane_ray_round<true, COUNT, NC, HK_POSTPONE_ANYHIT != 0, false, QN>(r, have && r.cur != DONE, sc, stack, lane, n_nodes, n_tris, cache, cursor < n || more);
#endif
    }
    stats += global_wave();
    HK_DBG_FLUSH(stats);
    wave_add(&stats->rays_shadow, n_casts);
    wave_add(&stats->hits, n_hits);
    if (COUNT) {
        wave_add(&stats->sh_nodes, n_nodes);
        wave_add(&stats->sh_tris, n_tris);
    }
}
template <bool COUNT, int STACK, int BLOCK = HK_TRACE_BLOCK, int NC = 0, bool QN = false>
__global__ void __launch_bounds__(BLOCK) k_shadow(DPathState st, DScene sc, int depth, DStats* stats) {
    __shared__ int lds_stack[(BLOCK / 64) * STACK * 64];
    __shared__ float4 lds_box[NC > 0 ? 3 * NC : 1];
    __shared__ int2 lds_child[NC > 0 ? NC : 1];
    int* stack = lds_stack + (threadIdx.x >> 6) * (STACK * 64);
    const NodeCache cache = node_cache_fill<NC, BLOCK, QN>(sc, lds_box, lds_child);
    shadow_body<COUNT, NC, QN>(st, sc, depth, stats, stream_open(st, ticket_ptr(st, depth, TK_SHADOW), false, depth, Q_SHADOW), stack, cache);
}

// K10 of bounce `depth` and K2 of bounce `depth + 1` of ONE segment in one per-lane-refill loop (k_small_pass): after the shade stage
// the segment's shadow rays and its next generation of rays are both there and depend on nothing but it, so idle lanes take whichever is
// left (shadow records first) and the wave keeps twice the rays in flight — one drain instead of two, fuller rounds.  Per ray nothing
// changes: an any-hit ray ends at its first accepted triangle and delivers its contribution (shadow_body), a closest-hit ray is routed
// as trace_lean_body routes it; a path's L sees the same additions in the same order (its shadow ray of this bounce, then whatever
// bounce depth + 1 adds in the NEXT stage).
template <int NC>
__device__ __forceinline__ void trace_shadow_body(const DPathState& st, const DScene& sc, int depth, DStats* stats, int gw, int* __restrict__ stack, const NodeCache& cache) {
    const int lane = lane_id();
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    const int DONE = (int)0x80000000;
    unsigned n_nodes = 0, n_tris = 0, n_casts = 0, n_hits = 0, n_sh_casts = 0;
    const int dt = depth + 1;
    const DPathGen g = st.gen[dt & 1];
    const uint32_t seg = (uint32_t)gw * (uint32_t)st.wave_cap;
    const int n_ray = *count_ptr(st, dt, Q_RAY, gw), n_sh = *count_ptr(st, depth, Q_SHADOW, gw);
    WaveQ q_escaped = wq_open(st.escaped_q, st, gw);
    int kind_count[HK_MAX_KINDS];
#pragma unroll
    for (int k = 0; k < HK_MAX_KINDS; ++k) kind_count[k] = 0;
    int cur_ray = 0, cur_sh = 0;
    int state = LR_EMPTY;
    uint32_t slot = 0;
    float rec_den = 0.0f;
    LaneRay r;
    r.cur = r.pend = DONE;
    r.any = false;
    for (;;) {
        const unsigned long long run_m = __ballot(state == LR_ACTIVE && r.cur != DONE);
        if (run_m == 0ull || (64 - __popcll(run_m) >= HK_TRACE_MIN_IDLE && (cur_ray < n_ray || cur_sh < n_sh))) {
            int kind = -1;
            if (state == LR_ACTIVE && r.cur == DONE) {
                if (r.any) {   // a shadow ray: unoccluded, it delivers its contribution
                    if (r.best.prim < 0) {
                        if (st.sh_final)
                            shadow_contribute_final(st, slot, rec_den);
                        else
                            shadow_contribute<true>(st, slot, s4(1.0f), s4(1.0f), s4(1.0f));
                    } else
                        ++n_hits;
                } else if (r.best.prim < 0)
                    kind = -2;
                else {
                    ++n_hits;
                    const DTriMeta hm = sc.meta[r.best.prim];
                    int mat = sc.mis[hm.mi].material;
                    if (sc.materials[mat].kind == HK_MAT_MIX) {
                        float w = 1.0f - r.best.u - r.best.v;
                        mat = resolve_mix_material(sc, mat, r.o + r.d * r.best.t, -r.d, uv_at(sc, r.best.prim, w, r.best.u, r.best.v));
                    }
                    kind = sc.materials[mat].kind;
                    if (kind == HK_MAT_MIX) kind = HK_MAT_FALLBACK;
                    stream_st(&st.hit[slot], make_float4(r.best.t, __int_as_float(r.best.prim), r.best.u, r.best.v));
                    stream_st(reinterpret_cast<uint32_t*>(st.mat_id) + slot, (uint32_t)(mat | (hm.arealight > 0 ? HK_MAT_EMISSIVE_BIT : 0)));
                }
                state = LR_EMPTY;
            }
            wq_push(q_escaped, slot, kind == -2);
            unsigned long long pending = __ballot(kind >= 0);
            while (pending) {
                int src = __ffsll((long long)pending) - 1;
                const int k = __builtin_amdgcn_readlane(kind, src);
                bool mine = kind == k;
                unsigned long long m = __ballot(mine);
                int cnt = 0;
#pragma unroll
                for (int kk = 0; kk < HK_MAX_KINDS; ++kk) cnt = (kk == k) ? kind_count[kk] : cnt;
                if (mine) stream_st(&st.mat_q[((size_t)k * st.n_waves + gw) * st.wave_cap + cnt + __popcll(m & lt_mask)], slot);
                int add = __popcll(m);
#pragma unroll
                for (int kk = 0; kk < HK_MAX_KINDS; ++kk) kind_count[kk] += (kk == k) ? add : 0;
                pending &= ~m;
            }
            // ---- refill: the shadow records first, then the rays of the next bounce ----
            const unsigned long long want = __ballot(state == LR_EMPTY);
            const int avail_sh = n_sh - cur_sh, avail_ray = n_ray - cur_ray;
            const int rank = __popcll(want & lt_mask);
            if (state == LR_EMPTY) {
                if (rank < avail_sh) {
                    slot = seg + (uint32_t)(cur_sh + rank);
                    float4 O = stream_ld(&st.sh_o[slot]), D = stream_ld(&st.sh_d[slot]);
                    rec_den = D.w;
                    if (O.w >= 1e-6f) {   // a degenerate shadow ray is simply not visible
                        ++n_sh_casts;
                        lane_ray_start(r, sc, mk3(O.x, O.y, O.z), mk3(D.x, D.y, D.z), O.w);
                        r.any = true;
                        state = LR_ACTIVE;
                    }
                } else if (rank - avail_sh < avail_ray) {
                    slot = seg + (uint32_t)(cur_ray + rank - avail_sh);
                    float4 O = stream_ld(&g.ray_o[slot]), D = stream_ld(&g.ray_d[slot]);
                    ++n_casts;
                    lane_ray_start(r, sc, mk3(O.x, O.y, O.z), mk3(D.x, D.y, D.z), O.w);
                    state = LR_ACTIVE;
                }
            }
            const int want_n = __popcll(want);
            const int take_sh = want_n < avail_sh ? want_n : avail_sh;
            const int rest = want_n - take_sh;
            cur_sh += take_sh;
            cur_ray += rest < avail_ray ? rest : avail_ray;
            if (__ballot(state == LR_ACTIVE) == 0ull) {
                if (cur_ray >= n_ray && cur_sh >= n_sh) break;
                continue;   // (only degenerate shadow rays were drawn: draw again)
            }
        }
        lane_ray_round<false, false, NC, false, true>(r, state == LR_ACTIVE && r.cur != DONE, sc, stack, lane, n_nodes, n_tris, cache, cur_ray < n_ray || cur_sh < n_sh);
    }
    wq_close(q_escaped, count_ptr(st, dt, Q_ESCAPED, gw));
    if (lane == 0) {
        *count_ptr(st, dt, Q_MEDIUM, gw) = 0;
#pragma unroll
        for (int k = 0; k < HK_MAX_KINDS; ++k) *count_ptr(st, dt, Q_MAT0 + k, gw) = kind_count[k];
    }
    stats += global_wave();
    wave_add(&stats->rays_closest, n_casts);
    wave_add(&stats->rays_shadow, n_sh_casts);
    wave_add(&stats->hits, n_hits);
}

// ---------------------------------------------------------------------------------------------------
// A SMALL pass as ONE launch (the reference's interactive call: one sample of every pixel, volpath.jl:445-450).  A path never leaves the
// segment that generated its camera ray, so K1 and every bounce's K2 / K8 - K9 / K10 of a segment depend on nothing outside it: the wave
// that owns the segment runs all of them, stage after stage, without waiting for any other wave.  As launches, the 26 kernels of such a
// call each ran at the latency of one wave's chunks PLUS the wait for the slowest wave of the stage; here a wave's stages follow each
// other directly.  The stage bodies are the standalone kernels' (camera_body, trace_lean_body, shade_body, shadow_body: same arithmetic,
// same order, same queues — films bit-identical), the block is k_trace_lean's (16 waves: the stacks and the top of the tree in LDS).
// Hand-over between stages goes through the segment's global arrays: written and read by the SAME wave, so completing the stores
// (workgroup-scope release / acquire: s_waitcnt; a CU's vector L1 is write-through and coherent with its own stores) is all it takes.
// For: all-opaque scenes without media, escape lights or light preselection whose only material kind is Matte under simple lights
// (launch_small_pass says which; everything else keeps the launches).
// ---------------------------------------------------------------------------------------------------
template <typename ACC>
__device__ __forceinline__ void film_tile(const DPathState& st, const DFrame& fr, const DTables& T, ACC* __restrict__ accum, float* __restrict__ mine, int tile);
__device__ __forceinline__ void stage_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
// GENERAL: the kinds with a light-weight shade body — Matte (under any lights), Mirror, Glass, Conductor — whichever the scene has, and K7 for
// escape lights (escaped_body); two waves per SIMD (512-thread blocks and below), where the union of their registers fits.
// STACK = 32: trees deeper than 16 levels (the 10^6-triangle scene; its next-event light comes from the shade body's own descent of the
// light BVH — what k_light_select would have chosen, test_light_preselection_is_result_neutral).
template <int NC, int BLOCK, bool GENERAL = false, int STACK = 16>
__global__ void __attribute__((amdgpu_flat_work_group_size(BLOCK, BLOCK), amdgpu_waves_per_eu(BLOCK / 256))) k_small_pass(DPathState st, DScene sc, DTables T, DFrame fr, DFilter flt, DCamera cam, DSobol sob,
                                                                                                                            int max_depth, int shadows, int merged, void* accum, int film_mode, uint32_t kinds_mask, DStats* stats) {
    __shared__ int lds_stack[(BLOCK / 64) * STACK * 64];
    __shared__ float4 lds_box[3 * NC];
    __shared__ int2 lds_child[NC];
    __shared__ uint32_t emit_list[(BLOCK / 64) * 128];
    int* stack = lds_stack + (threadIdx.x >> 6) * (STACK * 64);
    uint32_t* elist = emit_list + (threadIdx.x >> 6) * 128;
    const NodeCache cache = node_cache_fill<NC, BLOCK>(sc, lds_box, lds_child);
#ifdef HK_DEBUG_UTIL   // cycles per wave and stage (probes 10 - 13: camera, trace, shade, shadow — "x of y": average cycles of a wave's stage calls / 64)
    unsigned long long t_stage[4] = {0, 0, 0, 0}, n_stage[4] = {0, 0, 0, 0};
#define HK_STAGE_T(i, call) do { const unsigned long long t0_ = __builtin_readcyclecounter(); call; t_stage[i] += __builtin_readcyclecounter() - t0_; n_stage[i] += 64; } while (0)
#else
#define HK_STAGE_T(i, call) call
#endif
    for (int g = global_wave(); g < st.n_waves; g += physical_waves()) {
        HK_STAGE_T(0, camera_body(st, fr, T, flt, cam, sob, seg_single(st, g)));
        stage_fence();
        HK_STAGE_T(1, (trace_lean_body<false, NC>(st, sc, 0, stats, seg_single(st, g), stack, cache)));
        for (int depth = 0; depth < max_depth; ++depth) {
            stage_fence();
            if (!GENERAL)
                HK_STAGE_T(2, (shade_body<HK_MAT_MATTE, true, false, false>(st, sc, T, fr, sob, depth, 1, stats, seg_single(st, g), elist)));
            else {
                if (sc.has_escape_lights) escaped_body<1>(st, sc, T, depth, fr.implicit_ones, seg_single(st, g));
                int first_kind = 1;   // the kinds in ascending order, each continuing the shadow / next-ray queues of the one before (as the launches do)
#define HK_SMALL_KIND(K, S)                                                                                                                   \
    if (kinds_mask & (1u << K)) {                                                                                                             \
        if (!first_kind) stage_fence();                                                                                                       \
        HK_STAGE_T(2, (shade_body<K, S, false, false>(st, sc, T, fr, sob, depth, first_kind, stats, seg_single(st, g), elist)));            \
        first_kind = 0;                                                                                                                       \
    }
                if (sc.simple_lights) {
                    HK_SMALL_KIND(HK_MAT_MATTE, true)
                } else {
                    HK_SMALL_KIND(HK_MAT_MATTE, false)
                }
                HK_SMALL_KIND(HK_MAT_MIRROR, false)
                HK_SMALL_KIND(HK_MAT_GLASS, false)
                HK_SMALL_KIND(HK_MAT_CONDUCTOR, false)
#undef HK_SMALL_KIND
            }
            stage_fence();
            // the shadow rays of this bounce and the rays of the next one, together (trace_shadow_body)
            if (depth + 1 < max_depth) {
                if (shadows && merged)
                    HK_STAGE_T(3, (trace_shadow_body<NC>(st, sc, depth, stats, g, stack, cache)));
                else {
                    if (shadows) HK_STAGE_T(3, (shadow_body<false, NC>(st, sc, depth, stats, seg_single(st, g), stack, cache)));
                    HK_STAGE_T(1, (trace_lean_body<false, NC>(st, sc, depth + 1, stats, seg_single(st, g), stack, cache)));
                }
            } else if (shadows)
                HK_STAGE_T(3, (shadow_body<false, NC>(st, sc, depth, stats, seg_single(st, g), stack, cache)));
        }
        // K12 for a one-sample pass: chunk c of the camera IS tile c, and this segment's chunks are g, g + W, ... — the wave adds its own
        // paths to the film (film_tile: k_film's arithmetic; the traversal stacks are free now: 256 of their floats stage the colours)
        if (film_mode) {
            stage_fence();
            const int n_tiles = fr.n_pixels_padded >> 6;
            for (int tile = g; tile < n_tiles; tile += st.n_waves) {
                if (film_mode == 2)
                    film_tile<double>(st, fr, T, (double*)accum, reinterpret_cast<float*>(stack), tile);
                else
                    film_tile<float>(st, fr, T, (float*)accum, reinterpret_cast<float*>(stack), tile);
            }
        }
    }
#ifdef HK_DEBUG_UTIL
    if (lane_id() == 0)
        for (int i = 0; i < 4; ++i) {
            (stats + global_wave())->dbg[20 + 2 * i] += n_stage[i];
            (stats + global_wave())->dbg[21 + 2 * i] += t_stage[i];
        }
#endif
#undef HK_STAGE_T
}

#ifndef HK_GREY_WALK_WAVES
#define HK_GREY_WALK_WAVES 4
#endif
enum { SH_EMPTY = 0, SH_CAST = 1, SH_TRACK = 2 };
#ifndef HK_SHADOW_FEED_ROUNDS
#define HK_SHADOW_FEED_ROUNDS 3
#endif
#ifndef HK_SHADOW_TRACK_BATCH
#define HK_SHADOW_TRACK_BATCH 4
#endif

// GREY: see k_track — the transmittance ratios of a flat-spectrum medium have four equal components, so T_ray, r_u, r_l and the
// per-segment ratio-tracking state are one float each (same operations, in the same order, as the general code performs on the
// first component), no wavelengths are fetched, and a cell boundary costs nothing.
template <bool COUNT, int MM, int STACK = HK_LDS_STACK, bool GREY = false>
__global__ void __attribute__((amdgpu_flat_work_group_size(HK_TRACE_BLOCK, HK_TRACE_BLOCK), amdgpu_waves_per_eu(MM == 0 ? 4 : GREY ? HK_GREY_WALK_WAVES : (MM == 8 || MM == 2 || MM == 1) ? 3 : HK_MEDIA_WAVES))) k_shadow_walk(DPathState st, DScene sc, DTables T, int depth, int tune, DStats* stats, const DMedium* __restrict__ media) {
    __shared__ int lds_stack[(HK_TRACE_BLOCK / 64) * STACK * 64];
    int* stack = lds_stack + (threadIdx.x >> 6) * (STACK * 64);
    const int lane = lane_id();
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    unsigned n_nodes = 0, n_tris = 0, n_casts = 0, n_hits = 0, n_coll = 0, n_dda = 0;
    HK_DBG_DECL
    SegStream stream = stream_open(st, ticket_ptr(st, depth, TK_SHADOW), true, depth, Q_SHADOW);
    uint32_t rec0 = 0;   // current segment's shadow records: entries rec0 .. rec0 + n - 1 (the wave streams segment after segment)
    int n = 0, cursor = 0;
    bool more = true;
    int state = SH_EMPTY;
    uint32_t slot = 0;
    v3 ro = mk3(0, 0, 0), dir = mk3(0, 0, 1);
    float t_remaining = 0.0f, hit_t = 0.0f;
    int medium = -1, next_medium = -1, seg = 0;
    bool miss_case = false, transition = false;
    using W = typename std::conditional<GREY, float, S4>::type;   // a spectral weight: four wavelengths, or one value for all four
    W T_ray = wone<W>(), tr_u = wone<W>(), tr_l = wone<W>();
    S4 lambda = s4(0.0f);
    // ratio-tracking state of the current medium segment
    W sT = wone<W>(), su = wone<W>(), sl = wone<W>();
    S4 base_a = s4(0.0f), base_s = s4(0.0f), base_Le = s4(0.0f), sm = s4(0.0f);
    float a0 = 0.0f, s0 = 0.0f;   // GREY: the flat sigma_a / sigma_s values
    MajorantIter it = exhausted_iter();
    PCG32 rng = PCG32{0ull, 0ull};
    float seg1 = 0.0f, sm0 = 0.0f, t = 0.0f;
    bool in_seg = false, after_inner = false, pending = false;
    float pend_dt = 0.0f;
    int k_in_seg = 0, segi = 0;
    int feed_rounds = 0;
    for (;;) {
        // ---- refill ----
        const unsigned long long busy_m = __ballot(state != SH_EMPTY);
        if (busy_m == 0ull || (64 - __popcll(busy_m) >= (tune >> 24) && (cursor < n || more))) {
            while (more && cursor >= n) {   // this segment is used up: go on with the next one, the lanes in flight keep running
                const int gw = stream_next(stream, st.n_waves);
                if (gw >= st.n_waves) {
                    more = false;
                    break;
                }
                rec0 = (uint32_t)gw * (uint32_t)st.wave_cap;
                n = *count_ptr(st, depth, Q_SHADOW, gw);
                cursor = 0;
            }
            const unsigned long long want = ~busy_m;
            const int avail = n - cursor;
            const int rank = __popcll(want & lt_mask);
            if (state == SH_EMPTY && rank < avail) {
                slot = rec0 + (uint32_t)(cursor + rank);
                float4 O = st.sh_o[slot], D = st.sh_d[slot];
                ro = mk3(O.x, O.y, O.z);
                dir = mk3(D.x, D.y, D.z);
                t_remaining = O.w;
                medium = __float_as_int(D.w);
                T_ray = wone<W>();
                tr_u = wone<W>();
                tr_l = wone<W>();
                if (MM != 0 && !GREY) lambda = ld4(&st.lambda_s[st.sh_slot[slot]]);
                seg = 0;
                state = t_remaining < 1e-6f ? SH_EMPTY : SH_CAST;  // a degenerate ray is simply not visible
            }
            const int want_n = __popcll(want);
            cursor += want_n < avail ? want_n : (avail > 0 ? avail : 0);
            if (__ballot(state != SH_EMPTY) == 0ull) {
                if (cursor >= n && !more) break;
                continue;  // every fetched ray was degenerate (or the segment was empty): fetch again
            }
        }
        // ---- casts, for all lanes that need one (when enough of them wait, or nothing else can run) ----
        {
            const unsigned long long cast_m = __ballot(state == SH_CAST);
            const unsigned long long track_m = __ballot(state == SH_TRACK);
            if (cast_m != 0ull && (track_m == 0ull || __popcll(cast_m) >= ((tune >> 24) < 8 ? (tune >> 24) : 8))) {
                HK_DBG(0, state == SH_CAST);
                if (state == SH_CAST) {
                    bool opaque;
                    ++n_casts;
                    HitRec h = traverse<1, COUNT>(sc, ro, dir, t_remaining, stack, lane, n_nodes, n_tris, opaque);
                    bool alive = true;
                    if (h.prim < 0) {
                        miss_case = true;
                        hit_t = t_remaining;
                    } else {
                        ++n_hits;
                        miss_case = false;
                        hit_t = h.t;
                        if (opaque)
                            alive = false;
                        else {
                            DTriMeta meta = sc.meta[h.prim];
                            DMediumInterface mi = sc.mis[meta.mi];
                            v3 ng = geometric_normal(sc, h.prim);
                            bool entering = dot(dir, ng) < 0.0f;
                            transition = mi.inside != mi.outside;
                            next_medium = transition ? (entering ? mi.inside : mi.outside) : medium;
                            if (!transition) {
                                float w = 1.0f - h.u - h.v;
                                float alpha = surface_alpha(sc, mi.material, uv_at(sc, h.prim, w, h.u, h.v));
                                bool pass = false;
                                if (alpha < 1.0f) {
                                    PCG32 arng = pcg32_init(pbrt_hash(ro), pbrt_hash(dir));
                                    pass = pcg32_f32(arng) > alpha;
                                }
                                alive = pass;
                            }
                        }
                    }
                    if (!alive)
                        state = SH_EMPTY;  // blocked
                    else if (MM != 0 && medium >= 0) {
                        // ratio tracking over [0, hit_t] of this segment (intersection.jl:326-336, 376-386)
                        const DMedium& m = sc.n_media == 1 ? media[0] : media[medium];
                        sT = wone<W>();
                        su = wone<W>();
                        sl = wone<W>();
                        if constexpr (GREY) {
                            a0 = eval_flat(m.sigma_a);
                            s0 = eval_flat(m.sigma_s);
                            base_a = s4(a0);
                            base_s = s4(s0);
                        } else {
                            base_a = eval_scaled(m.sigma_a, lambda);
                            base_s = eval_scaled(m.sigma_s, lambda);
                            if (HK_HAS_MEDIUM(MM, HK_MEDIUM_HOMOGENEOUS)) base_Le = eval_scaled(m.Le, lambda);
                        }
                        it = create_majorant_iterator<MM>(m, ro, dir, hit_t);
                        rng = pcg32_init(pbrt_hash(ro), pbrt_hash(dir));
                        in_seg = false;
                        after_inner = false;
                        pending = false;
                        segi = 0;
                        state = SH_TRACK;
                    } else if (miss_case) {
                        shadow_contribute<MM == 0>(st, slot, wide(T_ray), wide(tr_u), wide(tr_l));
                        state = SH_EMPTY;
                    } else {
                        // step over the surface (no medium on this side)
                        bool stop = false;
                        if (transition) {
                            if (is_black(T_ray)) stop = true;
                            medium = next_medium;
                        }
                        ro = ro + dir * (hit_t + 1e-4f);
                        t_remaining = t_remaining - hit_t - 1e-4f;
                        ++seg;
                        state = (stop || seg >= 10 || t_remaining < 1e-6f) ? SH_EMPTY : SH_CAST;
                    }
                }
            }
        }
        // ---- short rays (one cast and done: shadow rays that start outside the medium, blocked ones) leave their lanes empty for the
        //      whole tracking batch below, which is where the time goes: fetch and cast again, a few times, until the wave is mostly
        //      tracking (the cloud's walk ran its collision rounds with 28 % of the lanes EMPTY although work was waiting) ----
        if (MM != 0 && feed_rounds < ((tune >> 16) & 0xff)) {
            const int waiting = __popcll(__ballot(state == SH_EMPTY)) + __popcll(__ballot(state == SH_CAST));
            if (waiting >= (tune >> 24) && (cursor < n || more)) {
                ++feed_rounds;
                continue;
            }
        }
        feed_rounds = 0;
        // ---- a few ratio-tracking rounds: cheap steps until a tentative collision is pending, then the collisions.  Lanes are
        //      served medium by medium so that the medium record is read through a wave-uniform index (scalar loads) ----
        if (MM != 0) {
            unsigned long long todo = __ballot(state == SH_TRACK);
            while (todo) {
                const int m_uniform = __builtin_amdgcn_readlane(medium, __ffsll((long long)todo) - 1);
                const bool mine = state == SH_TRACK && medium == m_uniform;
                const DMedium& med = media[m_uniform];
#pragma unroll 1
                for (int batch = 0; batch < (tune & 0xff); ++batch) {
                    if (__ballot(mine && state == SH_TRACK) == 0ull) break;
                    bool track_done = false;
#pragma unroll 1
                    for (int adv = 0; adv < ((tune >> 8) & 0xff); ++adv) {
                        const bool need = mine && state == SH_TRACK && !pending && !track_done;
                        if (__ballot(need) == 0ull) break;
                        HK_DBG(1, need);
                        if (!need) continue;
                        if (!in_seg) {
                            float seg0;
                            if (after_inner && is_black(sT))
                                track_done = true;
                            else if (segi >= 256 || !majorant_next<MM>(it, med, base_a + base_s, seg0, seg1, sm))
                                track_done = true;
                            else {
                                ++segi;
                                ++n_dda;
                                sm0 = sm.x;
                                if (sm0 >= 1e-10f) {
                                    t = seg0;
                                    in_seg = true;
                                    k_in_seg = 0;
                                }
                            }
                            after_inner = false;
                        }
                        if (!in_seg || track_done) {   // (a lane that has just entered a cell goes on in the same round: see k_track)
                        } else if (k_in_seg >= 100) {
                            in_seg = false;
                            after_inner = true;
                        } else {
                            ++k_in_seg;
                            float u = pcg32_f32(rng);
                            pend_dt = -media_logf(maxf(1e-10f, 1.0f - u)) / sm0;
                            float ts = t + pend_dt;
                            if (ts >= seg1) {
                                if constexpr (!GREY) {   // GREY: T_maj / T_maj[1] = 1, nothing changes
                                    float dr = seg1 - t;
                                    S4 Tm = s4exp((-dr) * sm);
                                    float T0 = Tm.x;
                                    if (T0 > 1e-10f) {
                                        sT = div4(sT * Tm, T0);
                                        sl = div4(sl * Tm, T0);
                                        su = div4(su * Tm, T0);
                                    }
                                }
                                in_seg = false;
                                after_inner = true;
                            } else
                                pending = true;
                        }
                    }
                    HK_DBG(2, mine && state == SH_TRACK && pending);
                    HK_DBG(3, state == SH_TRACK);
                    HK_DBG(4, state == SH_CAST);
                    HK_DBG(5, state == SH_EMPTY);
                    if (GREY && mine && state == SH_TRACK && pending) {
                        if constexpr (GREY) {
                            pending = false;
                            const float dt = pend_dt;
                            const float ts = t + dt;
                            ++n_coll;
                            const float d = sample_density<MM>(med, ro + dir * ts);
                            const float sn0 = maxf(sm0 - a0 * d - s0 * d, 0.0f);
                            const float Tm0 = media_expf((-dt) * sm0);
                            const float pr = Tm0 * sm0;
                            if (pr > 1e-10f) {
                                const float inv = 1.0f / pr;
                                sT = ((sT * Tm0) * sn0) * inv;
                                sl = ((sl * Tm0) * sm0) * inv;
                                su = ((su * Tm0) * sn0) * inv;
                                const float est = sT * (1.0f / maxf(1e-10f, average_flat(sl + su)));
                                if (est < 0.05f) {
                                    float rr = pcg32_f32(rng);
                                    if (rr < 0.75f) {
                                        sT = 0.0f;
                                        track_done = true;
                                    } else
                                        sT = sT / (1.0f - 0.75f);
                                }
                                if (sT == 0.0f) track_done = true;
                                t = ts;
                            } else {
                                sT = 0.0f;
                                track_done = true;
                            }
                        }
                    }
                    if (!GREY && mine && state == SH_TRACK && pending) {
                      if constexpr (!GREY) {
                        pending = false;
                        const float dt = pend_dt;
                        const float ts = t + dt;
                        ++n_coll;
                        MediumProps mp = sample_point<MM>(T, lambda, med, base_a, base_s, base_Le, ro + dir * ts);
                        S4 sn = s4max0(sm - mp.sigma_a - mp.sigma_s);
                        S4 Tm = s4exp((-dt) * sm);
                        float pr = Tm.x * sm0;
                        if (pr > 1e-10f) {
                            sT = div4(sT * Tm * sn, pr);
                            sl = div4(sl * Tm * sm, pr);
                            su = div4(su * Tm * sn, pr);
                            S4 est = div4(sT, maxf(1e-10f, average(sl + su)));
                            if (max_component(est) < 0.05f) {
                                float rr = pcg32_f32(rng);
                                if (rr < 0.75f) {
                                    sT = s4(0.0f);
                                    track_done = true;
                                } else
                                    sT = sT / (1.0f - 0.75f);
                            }
                            if (is_black(sT)) track_done = true;
                            t = ts;
                        } else {
                            sT = s4(0.0f);
                            track_done = true;
                        }
                    }
                    }
                    if (track_done) {
                        T_ray = T_ray * sT;
                        tr_u = tr_u * su;
                        tr_l = tr_l * sl;
                        if (miss_case) {
                            shadow_contribute<MM == 0>(st, slot, wide(T_ray), wide(tr_u), wide(tr_l));
                            state = SH_EMPTY;
                        } else {
                            bool stop = false;
                            if (transition) {
                                if (is_black(T_ray)) stop = true;
                                medium = next_medium;
                            }
                            ro = ro + dir * (hit_t + 1e-4f);
                            t_remaining = t_remaining - hit_t - 1e-4f;
                            ++seg;
                            state = (stop || seg >= 10 || t_remaining < 1e-6f) ? SH_EMPTY : SH_CAST;
                        }
                    }
                }
                todo &= ~__ballot(mine);
            }
        }
    }
    stats += global_wave();
    HK_DBG_FLUSH(stats);
    wave_add(&stats->sh_collisions, n_coll);
    wave_add(&stats->sh_nvdb_collisions, (sc.media_mask >> HK_MEDIUM_NANOVDB) & 1 ? n_coll : 0u);
    wave_add(&stats->sh_dda_steps, n_dda);
    wave_add(&stats->rays_shadow, n_casts);
    wave_add(&stats->hits, n_hits);
    if (COUNT) {
        wave_add(&stats->sh_nodes, n_nodes);
        wave_add(&stats->sh_tris, n_tris);
    }
}

// ---------------------------------------------------------------------------------------------------
// k_walk_pool: K10 of a scene whose ONE medium is GREY (k_shadow_walk<.., GREY = true> above; same arithmetic and RNG consumption per
// shadow ray, films bit-identical: tools/ab_bitwise.py, HK_WALK_POOL=0 runs the older kernel), with the two halves of a walk taken apart
// INSIDE the wave (the split into two kernels lost to the traffic and the random order of its hand-over queues):
//   * CAST PHASE, dense: whenever the wave's pool is empty and a lane is free, ALL 64 lanes cast — lanes that hold a ray waiting for its
//     next segment cast that one, every other lane (tracking or idle) the next shadow record of the stream — then run the surface logic
//     of intersection.jl:316-406 and, for the rays that have a stretch of medium in front of them, the set-up of the ratio tracker
//     (majorant iterator: 18 IEEE divisions; PCG seed: two 64-bit Murmur hashes).  What goes on is written to the POOL in LDS
//     (26 words per ray, CAP entries; a phase takes no more rays than the pool has room for).  In k_shadow_walk this code ran for
//     the 16 - 25 lanes that happened to be idle, and the tracking rounds ran with 46 % of the lanes waiting for it.
//     STACK / CAP: 16-entry traversal stacks and 48 pool entries are 36 KB of LDS per 4-wave block, four blocks per CU (8-entry stacks
//     and 64 pool entries for the example's shallow BVH measured the same: 0.305 against 0.308 s).
//   * TRACKING ROUNDS: a lane whose stretch ends takes the next ready ray from the pool in the same round (26 LDS reads); lane
//     flags in one VGPR, straight-line DDA step (see k_track_flat).
// ---------------------------------------------------------------------------------------------------
enum { WP_SLOT = 0, WP_RO = 1, WP_DIR = 4, WP_TREM = 7, WP_HIT = 8, WP_T = 9, WP_U = 10, WP_L = 11, WP_TMIN = 12, WP_TMAX = 13, WP_NT = 14, WP_DL = 17, WP_VOX = 20, WP_RNG = 21,
       WP_FL = 25, WP_FIELDS = 26 };
enum { WF_IN_SEG = 1, WF_PENDING = 2, WF_DONE = 4, WF_AFTER_INNER = 8, WF_IT_LIVE = 32, WF_NEG0 = 0x100, WF_NEG1 = 0x200, WF_NEG2 = 0x400, WF_MISS = 0x1000, WF_TRANSITION = 0x2000,
       WF_NEXT_MEDIUM = 0x4000, WF_MEDIUM = 0x8000, WF_SEG_SHIFT = 16, WF_SEG_MASK = 0xf0000, WF_CAST = 0x100000 };
#ifndef HK_WALK_POOL_WAVES
#define HK_WALK_POOL_WAVES 4
#endif
#ifndef HK_WALK_POOL_NC
#define HK_WALK_POOL_NC 16
#define HK_WALK_POOL_NT 32
#endif
template <bool COUNT, int MM, bool BRICKS, int STACK, int CAP>
__global__ void __attribute__((amdgpu_flat_work_group_size(HK_TRACE_BLOCK, HK_TRACE_BLOCK), amdgpu_waves_per_eu(HK_WALK_POOL_WAVES))) k_walk_pool(DPathState st, DScene sc, int depth, int tune, int gate, DStats* stats, const DMedium* __restrict__ media) {
    constexpr int WAVE_INTS = WP_FIELDS * CAP + STACK * 64;
    __shared__ int lds_all[(HK_TRACE_BLOCK / 64) * WAVE_INTS];
    int* const pool = lds_all + (threadIdx.x >> 6) * WAVE_INTS;   // [field][CAP]
    int* const stack = pool + WP_FIELDS * CAP;
    // the shallow-BVH instantiation (a cloud in a box: a few dozen triangles) casts from LDS: nodes and leaf triangles, 2.5 KB per block
    constexpr int NC = STACK <= 8 ? HK_WALK_POOL_NC : 0, NT = STACK <= 8 ? HK_WALK_POOL_NT : 0;
    __shared__ float4 lds_box[NC > 0 ? 3 * NC : 1];
    __shared__ int2 lds_child[NC > 0 ? NC : 1];
    __shared__ float4 lds_tri[NT > 0 ? 3 * NT : 1];
    const NodeCache cache = NC > 0 ? scene_cache_fill<NC, NT, HK_TRACE_BLOCK>(sc, lds_box, lds_child, lds_tri) : NodeCache();
    const int lane = lane_id();
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    unsigned n_nodes = 0, n_tris = 0, n_casts = 0, n_hits = 0, n_coll = 0, n_dda = 0;
    HK_DBG_DECL
    const DMedium& med = media[0];
    const float a0 = eval_flat(med.sigma_a), s0 = eval_flat(med.sigma_s);
    const float sig_t = a0 + s0;
    const int mrx = med.mres[0], mry = med.mres[1], mrz = med.mres[2];
    const float* __restrict__ maj = med.majorant;
    const int adv_n = (tune >> 8) & 0xff;
    const int gate_min = gate & 0xff, gate_cap = adv_n + ((gate >> 8) & 0xff);   // a collision round waits for gate_min pending lanes, for at most gate_cap cheap steps
    SegStream stream = stream_open(st, ticket_ptr(st, depth, TK_SHADOW), true, depth, Q_SHADOW);
    uint32_t rec0 = 0;   // current segment's shadow records: entries rec0 .. rec0 + n - 1 (the wave streams segment after segment)
    int n = 0, cursor = 0;
    bool more = true;
    int track_n = 0, cast_n = 0;   // wave-uniform: ready rays at pool[0 .. track_n), rays waiting for a cast at pool[CAP - cast_n .. CAP)
    int state = SH_EMPTY, fl = 0;
    uint32_t slot = 0;
    v3 ro = mk3(0, 0, 0), dir = mk3(0, 0, 1);
    float t_remaining = 0.0f, hit_t = 0.0f;
    float T_ray = 1.0f, tr_u = 1.0f, tr_l = 1.0f, sT = 1.0f, su = 1.0f, sl = 1.0f;
    float it_tmin = 0.0f, it_tmax = 0.0f, nt0 = 0.0f, nt1 = 0.0f, nt2 = 0.0f, dl0 = 0.0f, dl1 = 0.0f, dl2 = 0.0f;
    int vx = 0, vy = 0, vz = 0;
    PCG32 rng = PCG32{0ull, 0ull};
    float seg1 = 0.0f, sm0 = 0.0f, t = 0.0f, pend_dt = 0.0f;
    int k_in_seg = 0, segi = 0;
    for (;;) {
        const unsigned long long track_m = __ballot(state == SH_TRACK);
        // ---- cast phase: no ready ray is left in the pool and a lane is not tracking (or every lane holds a waiting ray) ----
        const bool own = state == SH_CAST;   // (a lane keeps a waiting ray itself only when the pool had no room for it)
        const unsigned long long own_m = __ballot(own);
        if (~track_m != 0ull && (track_n == 0 || own_m == ~0ull)) {
            // the rays of the phase: the lanes' own waiting ones, then the pool's waiting ones, then the next records of the stream — as
            // many of those as the pool could take back (every ray leaves at most one entry behind); the records may come from several
            // segments (the deep bounces' segments hold a few dozen records each)
            const int n_other = 64 - __popcll(own_m);
            const int c_take = cast_n < n_other ? cast_n : n_other;
            const int rank = __popcll(~own_m & lt_mask);
            const bool from_pool = !own && rank < c_take;
            bool fresh = false;
            uint32_t c_slot = slot;
            int f_take = 0;
            {
                int want = n_other - c_take < CAP - track_n - cast_n ? n_other - c_take : CAP - track_n - cast_n;
                while (want > 0) {
                    while (more && cursor >= n) {   // this segment is used up: go on with the next one
                        const int gw = stream_next(stream, st.n_waves);
                        if (gw >= st.n_waves) {
                            more = false;
                            break;
                        }
                        rec0 = (uint32_t)gw * (uint32_t)st.wave_cap;
                        n = *count_ptr(st, depth, Q_SHADOW, gw);
                        cursor = 0;
                    }
                    const int avail = n - cursor;
                    if (avail <= 0) break;
                    const int take = avail < want ? avail : want;
                    const int r = rank - c_take - f_take;
                    if (!own && r >= 0 && r < take) {
                        fresh = true;
                        c_slot = rec0 + (uint32_t)(cursor + r);
                    }
                    cursor += take;
                    f_take += take;
                    want -= take;
                }
            }
            if (own_m == 0ull && c_take == 0 && f_take == 0) {
                if (track_m == 0ull) break;   // nothing in flight, nothing left
            } else {
                v3 c_ro = ro, c_dir = dir;
                float c_trem = t_remaining, c_T = T_ray, c_u = tr_u, c_l = tr_l;
                int c_fl = fl & (WF_MEDIUM | WF_SEG_MASK);
                if (from_pool) {
                    const int* e = pool + (CAP - cast_n + rank);
                    c_slot = (uint32_t)e[WP_SLOT * CAP];
                    c_ro = mk3(__int_as_float(e[(WP_RO + 0) * CAP]), __int_as_float(e[(WP_RO + 1) * CAP]), __int_as_float(e[(WP_RO + 2) * CAP]));
                    c_dir = mk3(__int_as_float(e[(WP_DIR + 0) * CAP]), __int_as_float(e[(WP_DIR + 1) * CAP]), __int_as_float(e[(WP_DIR + 2) * CAP]));
                    c_trem = __int_as_float(e[WP_TREM * CAP]);
                    c_T = __int_as_float(e[WP_T * CAP]);
                    c_u = __int_as_float(e[WP_U * CAP]);
                    c_l = __int_as_float(e[WP_L * CAP]);
                    c_fl = e[WP_FL * CAP] & (WF_MEDIUM | WF_SEG_MASK);
                }
                if (fresh) {
                    const float4 O = st.sh_o[c_slot], D = st.sh_d[c_slot];
                    c_ro = mk3(O.x, O.y, O.z);
                    c_dir = mk3(D.x, D.y, D.z);
                    c_trem = O.w;
                    c_T = c_u = c_l = 1.0f;
                    c_fl = __float_as_int(D.w) >= 0 ? WF_MEDIUM : 0;
                }
                cast_n -= c_take;
                wave_lds_fence();
                const bool active = own || from_pool || (fresh && !(c_trem < 1e-6f));   // a degenerate ray is simply not visible
                int out = 0;   // 0: the ray ended here, 1: ready to track, 2: waits for another cast
                float c_hit = 0.0f;
                MajorantIter it = exhausted_iter();
                PCG32 c_rng = PCG32{0ull, 0ull};
                HK_DBG(8, active);
                if (active) {
                    bool opaque;
                    ++n_casts;
                    const HitRec h = traverse<1, COUNT, NC, NT>(sc, c_ro, c_dir, c_trem, stack, lane, n_nodes, n_tris, opaque, cache);
                    bool alive = true, miss_case = true, transition = false;
                    int next_medium = -1;
                    c_hit = c_trem;
                    if (h.prim >= 0) {
                        ++n_hits;
                        miss_case = false;
                        c_hit = h.t;
                        if (opaque)
                            alive = false;
                        else {
                            const DTriMeta meta = sc.meta[h.prim];
                            const DMediumInterface mi = sc.mis[meta.mi];
                            const v3 ng = geometric_normal(sc, h.prim);
                            const bool entering = dot(c_dir, ng) < 0.0f;
                            transition = mi.inside != mi.outside;
                            next_medium = transition ? (entering ? mi.inside : mi.outside) : ((c_fl & WF_MEDIUM) ? 0 : -1);
                            if (!transition) {
                                const float w = 1.0f - h.u - h.v;
                                const float alpha = surface_alpha(sc, mi.material, uv_at(sc, h.prim, w, h.u, h.v));
                                bool pass = false;
                                if (alpha < 1.0f) {
                                    PCG32 arng = pcg32_init(pbrt_hash(c_ro), pbrt_hash(c_dir));
                                    pass = pcg32_f32(arng) > alpha;
                                }
                                alive = pass;
                            }
                        }
                    }
                    if (!alive) {
                        // blocked
                    } else if (c_fl & WF_MEDIUM) {
                        // ratio tracking over [0, c_hit] of this segment (intersection.jl:326-336, 376-386): set up here, run in the tracking rounds
                        it = create_majorant_iterator<MM>(med, c_ro, c_dir, c_hit);
                        c_rng = pcg32_init(pbrt_hash(c_ro), pbrt_hash(c_dir));
                        c_fl |= (miss_case ? WF_MISS : 0) | (transition ? WF_TRANSITION : 0) | (next_medium >= 0 ? WF_NEXT_MEDIUM : 0) | ((it.mode & 0xff) == 2 ? WF_IT_LIVE : 0) | (it.mode & 0x700);
                        out = 1;
                    } else if (miss_case) {
                        shadow_contribute_rt(st, c_slot, s4(c_T), s4(c_u), s4(c_l));
                    } else {
                        // step over the surface (no medium on this side)
                        const bool stop = transition && c_T == 0.0f;
                        if (transition) c_fl = (c_fl & ~WF_MEDIUM) | (next_medium >= 0 ? WF_MEDIUM : 0);
                        c_ro = c_ro + c_dir * (c_hit + 1e-4f);
                        c_trem = c_trem - c_hit - 1e-4f;
                        const int seg = ((c_fl & WF_SEG_MASK) >> WF_SEG_SHIFT) + 1;
                        c_fl = (c_fl & ~WF_SEG_MASK) | (seg << WF_SEG_SHIFT);
                        out = (stop || seg >= 10 || c_trem < 1e-6f) ? 0 : 2;
                    }
                }
                HK_DBG(9, out == 1);
                const int c_vox = (c_fl & WF_IT_LIVE) ? (it.voxel[0] | (it.voxel[1] << 10) | (it.voxel[2] << 20)) : 0;
                // ---- a lane's own ray stays with the lane ----
                if (own) {
                    slot = c_slot;
                    ro = c_ro, dir = c_dir;
                    t_remaining = c_trem, hit_t = c_hit;
                    T_ray = c_T, tr_u = c_u, tr_l = c_l;
                    it_tmin = it.t_min, it_tmax = it.t_max;
                    nt0 = it.next_t[0], nt1 = it.next_t[1], nt2 = it.next_t[2];
                    dl0 = it.delta_t[0], dl1 = it.delta_t[1], dl2 = it.delta_t[2];
                    vx = c_vox & 1023, vy = (c_vox >> 10) & 1023, vz = c_vox >> 20;
                    rng = c_rng;
                    fl = c_fl;
                    sT = su = sl = 1.0f;
                    segi = 0;
                    state = out == 1 ? SH_TRACK : (out == 2 ? SH_CAST : SH_EMPTY);
                }
                // ---- the others' rays go to the pool: ready ones from the bottom, waiting ones from the top ----
                const unsigned long long ready_m = __ballot(!own && out == 1), wait_m = __ballot(!own && out == 2);
                if (!own && out != 0) {
                    int* e = pool + (out == 1 ? track_n + __popcll(ready_m & lt_mask) : CAP - 1 - cast_n - __popcll(wait_m & lt_mask));
                    e[WP_SLOT * CAP] = (int)c_slot;
                    e[(WP_RO + 0) * CAP] = __float_as_int(c_ro.x);
                    e[(WP_RO + 1) * CAP] = __float_as_int(c_ro.y);
                    e[(WP_RO + 2) * CAP] = __float_as_int(c_ro.z);
                    e[(WP_DIR + 0) * CAP] = __float_as_int(c_dir.x);
                    e[(WP_DIR + 1) * CAP] = __float_as_int(c_dir.y);
                    e[(WP_DIR + 2) * CAP] = __float_as_int(c_dir.z);
                    e[WP_TREM * CAP] = __float_as_int(c_trem);
                    e[WP_T * CAP] = __float_as_int(c_T);
                    e[WP_U * CAP] = __float_as_int(c_u);
                    e[WP_L * CAP] = __float_as_int(c_l);
                    e[WP_FL * CAP] = c_fl;
                    if (out == 1) {
                        e[WP_HIT * CAP] = __float_as_int(c_hit);
                        e[WP_TMIN * CAP] = __float_as_int(it.t_min);
                        e[WP_TMAX * CAP] = __float_as_int(it.t_max);
                        e[(WP_NT + 0) * CAP] = __float_as_int(it.next_t[0]);
                        e[(WP_NT + 1) * CAP] = __float_as_int(it.next_t[1]);
                        e[(WP_NT + 2) * CAP] = __float_as_int(it.next_t[2]);
                        e[(WP_DL + 0) * CAP] = __float_as_int(it.delta_t[0]);
                        e[(WP_DL + 1) * CAP] = __float_as_int(it.delta_t[1]);
                        e[(WP_DL + 2) * CAP] = __float_as_int(it.delta_t[2]);
                        e[WP_VOX * CAP] = c_vox;
                        e[(WP_RNG + 0) * CAP] = (int)(uint32_t)c_rng.state;
                        e[(WP_RNG + 1) * CAP] = (int)(uint32_t)(c_rng.state >> 32);
                        e[(WP_RNG + 2) * CAP] = (int)(uint32_t)c_rng.inc;
                        e[(WP_RNG + 3) * CAP] = (int)(uint32_t)(c_rng.inc >> 32);
                    }
                }
                track_n += __popcll(ready_m);
                cast_n += __popcll(wait_m);
                wave_lds_fence();
            }
        }
        // ---- free lanes take a ready ray from the pool ----
        {
            const unsigned long long free_m = __ballot(state == SH_EMPTY);
            if (free_m != 0ull && track_n > 0) {
                const int rank = __popcll(free_m & lt_mask);
                if (state == SH_EMPTY && rank < track_n) {
                    const int* e = pool + (track_n - 1 - rank);
                    slot = (uint32_t)e[WP_SLOT * CAP];
                    ro = mk3(__int_as_float(e[(WP_RO + 0) * CAP]), __int_as_float(e[(WP_RO + 1) * CAP]), __int_as_float(e[(WP_RO + 2) * CAP]));
                    dir = mk3(__int_as_float(e[(WP_DIR + 0) * CAP]), __int_as_float(e[(WP_DIR + 1) * CAP]), __int_as_float(e[(WP_DIR + 2) * CAP]));
                    t_remaining = __int_as_float(e[WP_TREM * CAP]);
                    hit_t = __int_as_float(e[WP_HIT * CAP]);
                    T_ray = __int_as_float(e[WP_T * CAP]);
                    tr_u = __int_as_float(e[WP_U * CAP]);
                    tr_l = __int_as_float(e[WP_L * CAP]);
                    it_tmin = __int_as_float(e[WP_TMIN * CAP]);
                    it_tmax = __int_as_float(e[WP_TMAX * CAP]);
                    nt0 = __int_as_float(e[(WP_NT + 0) * CAP]), nt1 = __int_as_float(e[(WP_NT + 1) * CAP]), nt2 = __int_as_float(e[(WP_NT + 2) * CAP]);
                    dl0 = __int_as_float(e[(WP_DL + 0) * CAP]), dl1 = __int_as_float(e[(WP_DL + 1) * CAP]), dl2 = __int_as_float(e[(WP_DL + 2) * CAP]);
                    const int vox = e[WP_VOX * CAP];
                    vx = vox & 1023, vy = (vox >> 10) & 1023, vz = vox >> 20;
                    rng.state = (uint64_t)(uint32_t)e[(WP_RNG + 0) * CAP] | ((uint64_t)(uint32_t)e[(WP_RNG + 1) * CAP] << 32);
                    rng.inc = (uint64_t)(uint32_t)e[(WP_RNG + 2) * CAP] | ((uint64_t)(uint32_t)e[(WP_RNG + 3) * CAP] << 32);
                    fl = e[WP_FL * CAP];
                    sT = su = sl = 1.0f;
                    segi = 0;
                    state = SH_TRACK;
                }
                const int free_n = __popcll(free_m);
                track_n -= free_n < track_n ? free_n : track_n;
                wave_lds_fence();
            }
        }
        if (__ballot(state == SH_TRACK) == 0ull) continue;   // (the cast phase above ends the kernel when nothing is left)
        // ---- phase A: cheap steps (next majorant cell, free-flight sample) ----
#pragma unroll 1
        for (int adv = 0;; ++adv) {
            const bool need = state == SH_TRACK && (fl & (WF_PENDING | WF_DONE)) == 0;
            if (__ballot(need) == 0ull) break;
            if (adv >= adv_n) {
                if (adv >= gate_cap) break;
                if (__popcll(__ballot(state == SH_TRACK && (fl & WF_PENDING) != 0)) >= gate_min) break;
            }
            HK_DBG(10, need);
            if (need) {
                if ((fl & WF_IN_SEG) == 0) {
                    if (((fl & WF_AFTER_INNER) != 0 && sT == 0.0f) || segi >= 256 || (fl & WF_IT_LIVE) == 0 || it_tmin >= it_tmax)
                        fl |= WF_DONE;
                    else {
                        // majorant_next (media.jl:625-729) of a DDA iterator, straight-line
                        const bool lxy = nt0 < nt1, lxz = nt0 < nt2, lyz = nt1 < nt2;
                        const bool ax0 = lxy & lxz, ax1 = (!lxy) & lyz;   // axis 0, axis 1, else axis 2
                        const float nt = ax0 ? nt0 : (ax1 ? nt1 : nt2);
                        const float stm = minf(nt, it_tmax);
                        const float rho = maj[vx + mrx * (vy + mry * vz)];
                        const float seg0 = it_tmin;
                        seg1 = stm;
                        const bool neg = (fl & (ax0 ? WF_NEG0 : (ax1 ? WF_NEG1 : WF_NEG2))) != 0;
                        const int v = (ax0 ? vx : (ax1 ? vy : vz)) + (neg ? -1 : 1);
                        const int lim = neg ? -1 : (ax0 ? mrx : (ax1 ? mry : mrz));
                        const float s = nt + (ax0 ? dl0 : (ax1 ? dl1 : dl2));
                        vx = ax0 ? v : vx;
                        vy = ax1 ? v : vy;
                        vz = (ax0 | ax1) ? vz : v;
                        nt0 = ax0 ? s : nt0;
                        nt1 = ax1 ? s : nt1;
                        nt2 = (ax0 | ax1) ? nt2 : s;
                        const bool out_of_grid = v == lim;
                        fl = out_of_grid ? (fl & ~WF_IT_LIVE) : fl;
                        it_tmin = out_of_grid ? it_tmax : stm;
                        ++segi;
                        ++n_dda;
                        sm0 = sig_t * rho;
                        if (sm0 >= 1e-10f) {
                            t = seg0;
                            fl |= WF_IN_SEG;
                            k_in_seg = 0;
                        }
                    }
                    fl &= ~WF_AFTER_INNER;
                }
                // a lane that has just entered a cell draws its first free flight in the same round
                if ((fl & (WF_IN_SEG | WF_DONE)) == WF_IN_SEG) {
                    if (k_in_seg >= 100)
                        fl = (fl & ~WF_IN_SEG) | WF_AFTER_INNER;
                    else {
                        ++k_in_seg;
                        const float u = pcg32_f32(rng);
                        pend_dt = -media_logf(maxf(1e-10f, 1.0f - u)) / sm0;
                        const float ts = t + pend_dt;
                        fl = ts >= seg1 ? ((fl & ~WF_IN_SEG) | WF_AFTER_INNER) : (fl | WF_PENDING);   // leaves the cell (T_maj / T_maj[1] = 1: nothing else changes) / tentative collision
                    }
                }
            }
        }
        // ---- phase B: the tentative collisions ----
        HK_DBG(11, state == SH_TRACK && (fl & WF_PENDING) != 0);
        HK_DBG(12, state == SH_TRACK);
        HK_DBG(13, state == SH_CAST);
        if (state == SH_TRACK && (fl & WF_PENDING) != 0) {
            fl &= ~WF_PENDING;
            const float dt = pend_dt;
            const float ts = t + dt;
            ++n_coll;
            const float d = sample_density<MM, BRICKS>(med, ro + dir * ts);
            const float sn0 = maxf(sm0 - a0 * d - s0 * d, 0.0f);
            const float Tm0 = media_expf((-dt) * sm0);
            const float pr = Tm0 * sm0;
            if (pr > 1e-10f) {
                const float inv = 1.0f / pr;
                sT = ((sT * Tm0) * sn0) * inv;
                sl = ((sl * Tm0) * sm0) * inv;
                su = ((su * Tm0) * sn0) * inv;
                const float est = sT * (1.0f / maxf(1e-10f, average_flat(sl + su)));
                if (est < 0.05f) {
                    const float rr = pcg32_f32(rng);
                    if (rr < 0.75f) {
                        sT = 0.0f;
                        fl |= WF_DONE;
                    } else
                        sT = sT / (1.0f - 0.75f);
                }
                if (sT == 0.0f) fl |= WF_DONE;
                t = ts;
            } else {
                sT = 0.0f;
                fl |= WF_DONE;
            }
        }
        // ---- the end of a stretch of medium: contribute (the ray reached its light), or leave the ray in the pool for the cast behind the surface ----
        {
            const bool fin = state == SH_TRACK && (fl & WF_DONE) != 0;
            const unsigned long long fin_m = __ballot(fin);
            if (fin_m != 0ull) {
                bool goes_on = false;
                if (fin) {
                    T_ray = T_ray * sT;
                    tr_u = tr_u * su;
                    tr_l = tr_l * sl;
                    state = SH_EMPTY;
                    if (fl & WF_MISS)
                        shadow_contribute_rt(st, slot, s4(T_ray), s4(tr_u), s4(tr_l));
                    else {
                        const bool stop = (fl & WF_TRANSITION) != 0 && T_ray == 0.0f;
                        const int medium_bit = (fl & WF_TRANSITION) ? ((fl & WF_NEXT_MEDIUM) ? WF_MEDIUM : 0) : WF_MEDIUM;
                        ro = ro + dir * (hit_t + 1e-4f);
                        t_remaining = t_remaining - hit_t - 1e-4f;
                        const int seg = ((fl & WF_SEG_MASK) >> WF_SEG_SHIFT) + 1;
                        fl = medium_bit | (seg << WF_SEG_SHIFT);
                        goes_on = !(stop || seg >= 10 || t_remaining < 1e-6f);
                    }
                }
                const unsigned long long on_m = __ballot(goes_on);
                if (on_m != 0ull) {
                    const int room = CAP - track_n - cast_n;
                    const int r = __popcll(on_m & lt_mask);
                    if (goes_on && r < room) {
                        int* e = pool + (CAP - 1 - cast_n - r);
                        e[WP_SLOT * CAP] = (int)slot;
                        e[(WP_RO + 0) * CAP] = __float_as_int(ro.x);
                        e[(WP_RO + 1) * CAP] = __float_as_int(ro.y);
                        e[(WP_RO + 2) * CAP] = __float_as_int(ro.z);
                        e[(WP_DIR + 0) * CAP] = __float_as_int(dir.x);
                        e[(WP_DIR + 1) * CAP] = __float_as_int(dir.y);
                        e[(WP_DIR + 2) * CAP] = __float_as_int(dir.z);
                        e[WP_TREM * CAP] = __float_as_int(t_remaining);
                        e[WP_T * CAP] = __float_as_int(T_ray);
                        e[WP_U * CAP] = __float_as_int(tr_u);
                        e[WP_L * CAP] = __float_as_int(tr_l);
                        e[WP_FL * CAP] = fl;
                    } else if (goes_on)
                        state = SH_CAST;   // no room: the lane keeps the ray until the next cast phase
                    const int pushed = __popcll(on_m);
                    cast_n += pushed < room ? pushed : room;
                    wave_lds_fence();
                }
            }
        }
    }
    stats += global_wave();
    HK_DBG_FLUSH(stats);
    wave_add(&stats->sh_collisions, n_coll);
    wave_add(&stats->sh_nvdb_collisions, (sc.media_mask >> HK_MEDIUM_NANOVDB) & 1 ? n_coll : 0u);
    wave_add(&stats->sh_dda_steps, n_dda);
    wave_add(&stats->rays_shadow, n_casts);
    wave_add(&stats->hits, n_hits);
    if (COUNT) {
        wave_add(&stats->sh_nodes, n_nodes);
        wave_add(&stats->sh_tris, n_tris);
    }
}

// ---------------------------------------------------------------------------------------------------
// K10 of a scene whose (single) medium is GREY, SPLIT in two kernels.  k_shadow_walk carries the registers of the
// BVH traversal and of the ratio tracker together (128 VGPRs with spills at 4 waves per SIMD) and these loops are bound by instruction
// issue at low residency; the two halves never need each other's state:
//   k_walk_cast   one any-hit-style cast per record (LDS stacks + the scene's nodes and triangles in LDS, like k_trace), the surface
//                 logic of intersection.jl:316-406; a ray that has to cross a medium is parked in its record and queued for
//   k_walk_track  ratio tracking through the medium up to the surface found by the cast (intersection.jl:422-542): a flat state
//                 machine without traversal state; a ray that goes on behind the surface is queued for the next cast round.
// Rounds alternate cast(0), track(0), cast(1), ... on the host side (HK_WALK_ROUNDS casts: the reference walks <= 10 segments).
// Round 0 streams the shadow records of the wave segments like k_shadow_walk; later rounds read GLOBAL index queues filled with one
// atomic per wave and push: the order of the queue does not matter to the result, because a path slot receives at most one shadow
// contribution per depth (one NEE sample per vertex) — no two records of a launch add to the same L.
// In-flight state of a record: sh_o = (origin, t_remaining), sh_d = (direction, medium), sh_T = (T_ray, r_u, r_l, hit_t),
// sh_aux = segments | miss << 8 | transition << 9 | (next medium + 1) << 16.
// Same arithmetic per ray as k_shadow_walk<.., GREY = true>: films are bit-identical (tests: HK_WALK_SPLIT=0 against 1).
// ---------------------------------------------------------------------------------------------------
#define HK_WALK_ROUNDS 10
HKD int* walk_ctl(const DPathState& st, int depth, int round) { return st.wq_ctl + (size_t)(depth * (HK_WALK_ROUNDS + 1) + round) * 4; }   // {count A, cursor A, count B, cursor B}
// A global queue is written and read in wave-private CHUNKS: one atomic on the queue's count / cursor per chunk, not per push
// (the first version paid one atomic per refill round and push — 90 M same-address atomics per frame: the tracking half alone took
// longer than the whole unsplit walk).  A writer reserves HK_GQ_CHUNK entries at a time and pads what it leaves unused with
// HK_GQ_NONE, which readers skip; a reader takes chunks sized to the queue (n / (8 x waves), 64 ... 4096 entries).
#define HK_GQ_CHUNK 256
#define HK_GQ_NONE 0xffffffffu
struct GQOut {
    uint32_t* q;
    int* count;
    int base, used, cap;
};
HKD GQOut gq_out_open(uint32_t* q, int* count) { return GQOut{q, count, 0, 0, 0}; }
HKD void gq_out_pad(GQOut& o) {
    for (int i = o.used + lane_id(); i < o.cap; i += 64) o.q[o.base + i] = HK_GQ_NONE;
}
HKD void gq_out_push(GQOut& o, uint32_t value, bool active) {   // called by the whole wave
    const unsigned long long m = __ballot(active);
    const int k = __popcll(m);
    if (k == 0) return;
    if (o.used + k > o.cap) {
        gq_out_pad(o);
        int b = 0;
        if (lane_id() == 0) b = atomicAdd(o.count, HK_GQ_CHUNK);
        o.base = __builtin_amdgcn_readfirstlane(b);
        o.used = 0;
        o.cap = HK_GQ_CHUNK;
    }
    if (active) o.q[o.base + o.used + __popcll(m & ((1ull << lane_id()) - 1ull))] = value;
    o.used += k;
}
struct GQIn {
    const uint32_t* q;
    int* cursor;
    int n, lo, hi, chunk;
    bool more;
};
HKD GQIn gq_in_open(const uint32_t* q, int* cursor, int n) {
    int chunk = (n / (8 * physical_waves()) + 63) & ~63;
    chunk = chunk < 64 ? 64 : (chunk > 4096 ? 4096 : chunk);
    return GQIn{q, cursor, n, 0, 0, chunk, n > 0};
}
// the next entries of the wave's chunk for the lanes with `want` (lane rank r gets entry lo + r); -> false once the queue is used up
HKD bool gq_in_take(GQIn& in, bool want, uint32_t& value, bool& got) {
    got = false;
    while (in.lo >= in.hi && in.more) {
        int b = 0;
        if (lane_id() == 0) b = atomicAdd(in.cursor, in.chunk);
        b = __builtin_amdgcn_readfirstlane(b);
        if (b >= in.n)
            in.more = false;
        else {
            in.lo = b;
            in.hi = b + in.chunk < in.n ? b + in.chunk : in.n;
        }
    }
    if (in.lo >= in.hi) return false;
    const unsigned long long m = __ballot(want);
    const int rank = __popcll(m & ((1ull << lane_id()) - 1ull));
    const int avail = in.hi - in.lo, k = __popcll(m);
    if (want && rank < avail) {
        value = in.q[in.lo + rank];
        got = value != HK_GQ_NONE;
    }
    in.lo += k < avail ? k : avail;
    return true;
}

template <bool COUNT, int MM, int STACK>
__global__ void __launch_bounds__(HK_TRACE_BLOCK) k_walk_cast(DPathState st, DScene sc, int depth, int round, DStats* stats, const DMedium* __restrict__ media) {
    __shared__ int lds_stack[(HK_TRACE_BLOCK / 64) * STACK * 64];
    __shared__ float4 lds_box[3 * HK_MEDIA_NC];
    __shared__ int2 lds_child[HK_MEDIA_NC];
    __shared__ float4 lds_tri[3 * HK_MEDIA_NT];
    int* stack = lds_stack + (threadIdx.x >> 6) * (STACK * 64);
    int* ctl = walk_ctl(st, depth, round);
    if (round > 0 && ctl[0] == 0) return;   // nothing was queued for this round (uniform: before the block's barrier)
    const NodeCache cache = scene_cache_fill<HK_MEDIA_NC, HK_MEDIA_NT, HK_TRACE_BLOCK>(sc, lds_box, lds_child, lds_tri);
    const int lane = lane_id();
    unsigned n_nodes = 0, n_tris = 0, n_casts = 0, n_hits = 0;
    SegStream stream = stream_open(st, ticket_ptr(st, depth, TK_SHADOW), true, depth, Q_SHADOW);
    uint32_t rec0 = 0;
    int n = 0, cursor = 0;
    GQIn in = gq_in_open(st.wq_a, &ctl[1], round > 0 ? ctl[0] : 0);
    GQOut out = gq_out_open(st.wq_b, &ctl[2]);
    for (;;) {
        // ---- the next 64 records ----
        uint32_t rec = 0;
        bool have = false;
        if (round == 0) {
            if (cursor >= n) {
                const int gw = stream_next(stream, st.n_waves);
                if (gw >= st.n_waves) break;
                rec0 = (uint32_t)gw * (uint32_t)st.wave_cap;
                n = *count_ptr(st, depth, Q_SHADOW, gw);
                cursor = 0;
                continue;
            }
            have = cursor + lane < n;
            rec = rec0 + (uint32_t)(cursor + lane);
            cursor += 64;
        } else if (!gq_in_take(in, true, rec, have))
            break;
        v3 ro = mk3(0, 0, 0), dir = mk3(0, 0, 1);
        float t_remaining = 0.0f, T_ray = 1.0f, tr_u = 1.0f, tr_l = 1.0f;
        int medium = -1, seg = 0;
        if (have) {
            const float4 O = st.sh_o[rec], D = st.sh_d[rec];
            ro = mk3(O.x, O.y, O.z);
            dir = mk3(D.x, D.y, D.z);
            t_remaining = O.w;
            medium = __float_as_int(D.w);
            if (round > 0) {
                const float4 Tq = st.sh_T[rec];
                T_ray = Tq.x, tr_u = Tq.y, tr_l = Tq.z;
                seg = (int)(st.sh_aux[rec] & 0xffu);
            }
        }
        bool active = have && !(t_remaining < 1e-6f);   // a degenerate ray is simply not visible
        // ---- casts; a ray that meets a surface outside every medium steps over it and casts again right here ----
        while (__ballot(active) != 0ull) {
            bool park = false;
            if (active) {
                bool opaque;
                ++n_casts;
                HitRec h = traverse<1, COUNT, HK_MEDIA_NC, HK_MEDIA_NT>(sc, ro, dir, t_remaining, stack, lane, n_nodes, n_tris, opaque, cache);
                bool alive = true, miss_case = true, transition = false;
                int next_medium = medium;
                float hit_t = t_remaining;
                if (h.prim >= 0) {
                    ++n_hits;
                    miss_case = false;
                    hit_t = h.t;
                    if (opaque)
                        alive = false;
                    else {
                        DTriMeta meta = sc.meta[h.prim];
                        DMediumInterface mi = sc.mis[meta.mi];
                        v3 ng = geometric_normal(sc, h.prim);
                        bool entering = dot(dir, ng) < 0.0f;
                        transition = mi.inside != mi.outside;
                        next_medium = transition ? (entering ? mi.inside : mi.outside) : medium;
                        if (!transition) {
                            float w = 1.0f - h.u - h.v;
                            float alpha = surface_alpha(sc, mi.material, uv_at(sc, h.prim, w, h.u, h.v));
                            bool pass = false;
                            if (alpha < 1.0f) {
                                PCG32 arng = pcg32_init(pbrt_hash(ro), pbrt_hash(dir));
                                pass = pcg32_f32(arng) > alpha;
                            }
                            alive = pass;
                        }
                    }
                }
                if (!alive)
                    active = false;  // blocked
                else if (medium >= 0) {
                    // ratio tracking over [0, hit_t] comes next: park the ray in its record
                    st.sh_o[rec] = make_float4(ro.x, ro.y, ro.z, t_remaining);
                    st.sh_d[rec] = make_float4(dir.x, dir.y, dir.z, __int_as_float(medium));
                    st.sh_T[rec] = make_float4(T_ray, tr_u, tr_l, hit_t);
                    st.sh_aux[rec] = (uint32_t)seg | (miss_case ? 0x100u : 0u) | (transition ? 0x200u : 0u) | ((uint32_t)(next_medium + 1) << 16);
                    // the tracker's per-ray set-up (majorant iterator: three divisions per axis; two 64-bit hashes for the PCG32 seed) is
                    // done HERE, where every lane of the wave has a ray, and handed over in the record: inside the tracking state
                    // machine it ran on the ~25 lanes of a refill round and cost as much as five collisions
                    const MajorantIter it = create_majorant_iterator<MM>(media[0], ro, dir, hit_t);
                    const PCG32 rng = pcg32_init(pbrt_hash(ro), pbrt_hash(dir));
                    float4* itp = st.sh_it + 4 * (size_t)rec;
                    itp[0] = make_float4(it.next_t[0], it.next_t[1], it.next_t[2], it.t_min);
                    itp[1] = make_float4(it.delta_t[0], it.delta_t[1], it.delta_t[2], it.t_max);
                    itp[2] = make_float4(__int_as_float(it.voxel[0]), __int_as_float(it.voxel[1]), __int_as_float(it.voxel[2]), __int_as_float(it.mode));
                    itp[3] = make_float4(__uint_as_float((uint32_t)rng.state), __uint_as_float((uint32_t)(rng.state >> 32)), __uint_as_float((uint32_t)rng.inc),
                                         __uint_as_float((uint32_t)(rng.inc >> 32)));
                    park = true;
                    active = false;
                } else if (miss_case) {
                    shadow_contribute<false>(st, rec, s4(T_ray), s4(tr_u), s4(tr_l));
                    active = false;
                } else {
                    // step over the surface (no medium on this side)
                    bool stop = false;
                    if (transition) {
                        if (T_ray == 0.0f) stop = true;
                        medium = next_medium;
                    }
                    ro = ro + dir * (hit_t + 1e-4f);
                    t_remaining = t_remaining - hit_t - 1e-4f;
                    ++seg;
                    active = !(stop || seg >= 10 || t_remaining < 1e-6f);
                }
            }
            gq_out_push(out, rec, park);
        }
    }
    gq_out_pad(out);
    stats += global_wave();
    wave_add(&stats->rays_shadow, n_casts);
    wave_add(&stats->hits, n_hits);
    if (COUNT) {
        wave_add(&stats->sh_nodes, n_nodes);
        wave_add(&stats->sh_tris, n_tris);
    }
}

#ifndef HK_WALK_TRACK_WAVES
#define HK_WALK_TRACK_WAVES 5
#endif
template <int MM>
__global__ void __attribute__((amdgpu_flat_work_group_size(256, 256), amdgpu_waves_per_eu(HK_WALK_TRACK_WAVES))) k_walk_track(DPathState st, DScene sc, int depth, int round, int tune, DStats* stats,
                                                                                                                               const DMedium* __restrict__ media) {
    int* ctl = walk_ctl(st, depth, round);
    int* ctl_next = walk_ctl(st, depth, round + 1);
    const int n = ctl[2];
    if (n == 0) return;
    const DMedium& med = media[0];   // GREY scenes hold one medium
    const float a0 = eval_flat(med.sigma_a), s0 = eval_flat(med.sigma_s);
    const S4 sigma_t = s4(a0 + s0);
    const int refill_idle = tune >> 24, adv_rounds = (tune >> 8) & 0xff, batches = tune & 0xff;
    unsigned n_coll = 0, n_dda = 0;
    HK_DBG_DECL
    GQIn in = gq_in_open(st.wq_b, &ctl[3], n);
    GQOut out = gq_out_open(st.wq_a, &ctl_next[0]);
    // per-lane state
    bool busy = false, in_seg = false, pending = false;
    uint32_t rec = 0;
    v3 ro = mk3(0, 0, 0), dir = mk3(0, 0, 1);
    float T_ray = 1.0f, tr_u = 1.0f, tr_l = 1.0f, hit_t = 0.0f, t_remaining = 0.0f;
    float sT = 1.0f, su = 1.0f, sl = 1.0f, sm0 = 0.0f, seg1 = 0.0f, t = 0.0f, pend_dt = 0.0f;
    uint32_t aux = 0;
    int k_in_seg = 0, segi = 0;
    MajorantIter it = exhausted_iter();
    PCG32 rng = PCG32{0ull, 0ull};
    for (;;) {
        // ---- refill ----
        const unsigned long long busy_m = __ballot(busy);
        const bool can_refill = in.more || in.lo < in.hi;
        if (busy_m == 0ull || (can_refill && 64 - __popcll(busy_m) >= refill_idle)) {
            if (!can_refill) break;
            uint32_t r_new = 0;
            bool got = false;
            gq_in_take(in, !busy, r_new, got);
            HK_DBG(5, got);
            if (got) {
                rec = r_new;
                const float4 O = st.sh_o[rec], D = st.sh_d[rec], Tq = st.sh_T[rec];
                aux = st.sh_aux[rec];
                ro = mk3(O.x, O.y, O.z);
                dir = mk3(D.x, D.y, D.z);
                t_remaining = O.w;
                T_ray = Tq.x, tr_u = Tq.y, tr_l = Tq.z, hit_t = Tq.w;
                sT = su = sl = 1.0f;
                {   // iterator and RNG as k_walk_cast set them up
                    const float4* itp = st.sh_it + 4 * (size_t)rec;
                    const float4 I0 = itp[0], I1 = itp[1], I2 = itp[2], I3 = itp[3];
                    it.next_t[0] = I0.x, it.next_t[1] = I0.y, it.next_t[2] = I0.z, it.t_min = I0.w;
                    it.delta_t[0] = I1.x, it.delta_t[1] = I1.y, it.delta_t[2] = I1.z, it.t_max = I1.w;
                    it.voxel[0] = __float_as_int(I2.x), it.voxel[1] = __float_as_int(I2.y), it.voxel[2] = __float_as_int(I2.z), it.mode = __float_as_int(I2.w);
                    rng.state = (uint64_t)__float_as_uint(I3.x) | ((uint64_t)__float_as_uint(I3.y) << 32);
                    rng.inc = (uint64_t)__float_as_uint(I3.z) | ((uint64_t)__float_as_uint(I3.w) << 32);
                }
                in_seg = false;
                pending = false;
                segi = 0;
                busy = true;
            }
            if (__ballot(busy) == 0ull) {
                if (!(in.more || in.lo < in.hi)) break;
                continue;
            }
        }
#pragma unroll 1
        for (int batch = 0; batch < batches; ++batch) {
            bool done = false;
            // ---- cheap steps until a tentative collision is pending: next majorant cell; free-flight sample inside a cell ----
#pragma unroll 1
            for (int adv = 0; adv < adv_rounds; ++adv) {
                const bool need = busy && !pending && !done;
                if (__ballot(need) == 0ull) break;
                HK_DBG(0, need);
                HK_DBG(1, need && !in_seg);
                HK_DBG(2, need && in_seg);
                if (need && !in_seg) {
                    float seg0;
                    S4 sm;
                    if (segi >= 256 || !majorant_next<MM>(it, med, sigma_t, seg0, seg1, sm))
                        done = true;
                    else {
                        ++segi;
                        ++n_dda;
                        sm0 = sm.x;
                        const bool enter = sm0 >= 1e-10f;
                        t = enter ? seg0 : t;
                        in_seg = enter;
                        k_in_seg = enter ? 0 : k_in_seg;
                    }
                }
                if (need && in_seg && !done) {   // in a cell with a non-zero majorant (also: just entered): the next tentative collision, or out through the cell's far side
                    const bool over = k_in_seg >= 100;
                    PCG32 r2 = rng;
                    const float u = pcg32_f32(r2);
                    rng = over ? rng : r2;
                    k_in_seg += over ? 0 : 1;
                    pend_dt = -media_logf(maxf(1e-10f, 1.0f - u)) / sm0;
                    const bool leave = over || (t + pend_dt >= seg1);   // T_maj / T_maj[1] = 1 at the boundary: nothing else changes
                    in_seg = !leave;
                    pending = !leave;
                }
            }
            // ---- the tentative collisions ----
            HK_DBG(3, busy && pending);
            HK_DBG(4, busy);
            if (busy && pending) {
                pending = false;
                const float dt = pend_dt;
                const float ts = t + dt;
                ++n_coll;
                const float d = sample_density<MM>(med, ro + dir * ts);
                const float sn0 = maxf(sm0 - a0 * d - s0 * d, 0.0f);
                const float Tm0 = media_expf((-dt) * sm0);
                const float pr = Tm0 * sm0;
                if (pr > 1e-10f) {
                    const float inv = 1.0f / pr;
                    sT = ((sT * Tm0) * sn0) * inv;
                    sl = ((sl * Tm0) * sm0) * inv;
                    su = ((su * Tm0) * sn0) * inv;
                    const float est = sT * (1.0f / maxf(1e-10f, average_flat(sl + su)));
                    if (est < 0.05f) {
                        const float rr = pcg32_f32(rng);
                        if (rr < 0.75f) {
                            sT = 0.0f;
                            done = true;
                        } else
                            sT = sT / (1.0f - 0.75f);
                    }
                    if (sT == 0.0f) done = true;
                    t = ts;
                } else {
                    sT = 0.0f;
                    done = true;
                }
            }
            // ---- the end of this medium segment ----
            bool requeue = false;
            if (busy && done) {
                busy = false;
                T_ray = T_ray * sT;
                tr_u = tr_u * su;
                tr_l = tr_l * sl;
                if (aux & 0x100u)   // the cast ended at the light: deliver
                    shadow_contribute<false>(st, rec, s4(T_ray), s4(tr_u), s4(tr_l));
                else {
                    int seg = (int)(aux & 0xffu);
                    int medium = 0;   // the ray was in the (only) medium
                    bool stop = false;
                    if (aux & 0x200u) {
                        if (T_ray == 0.0f) stop = true;
                        medium = (int)(aux >> 16) - 1;
                    }
                    ro = ro + dir * (hit_t + 1e-4f);
                    t_remaining = t_remaining - hit_t - 1e-4f;
                    ++seg;
                    if (!(stop || seg >= 10 || t_remaining < 1e-6f)) {
                        st.sh_o[rec] = make_float4(ro.x, ro.y, ro.z, t_remaining);
                        st.sh_d[rec] = make_float4(dir.x, dir.y, dir.z, __int_as_float(medium));
                        st.sh_T[rec] = make_float4(T_ray, tr_u, tr_l, 0.0f);
                        st.sh_aux[rec] = (uint32_t)seg;
                        requeue = true;
                    }
                }
            }
            gq_out_push(out, rec, requeue);
            if (__ballot(busy) == 0ull) break;
        }
    }
    gq_out_pad(out);
    stats += global_wave();
    HK_DBG_FLUSH(stats);
    wave_add(&stats->sh_collisions, n_coll);
    wave_add(&stats->sh_nvdb_collisions, (sc.media_mask >> HK_MEDIUM_NANOVDB) & 1 ? n_coll : 0u);
    wave_add(&stats->sh_dda_steps, n_dda);
}

// ---------------------------------------------------------------------------------------------------
// K12: spectral -> RGB, firefly clamp, filter-weighted accumulation (volpath.jl:326-375).  The S samples of
// a pixel are folded in sample order, so the fp32 sums equal the reference's sample-by-sample sums.
// ---------------------------------------------------------------------------------------------------
// The wavelength pdfs are a function of the wavelengths alone (sample_wavelengths_visible: pdf = visible_wavelengths_pdf(lambda), the
// same expression on the same bits): the film kernel recomputes them instead of reading 16 B per sample that k_camera had to write.
HKD S4 pdf_of(S4 l) { return s4(visible_wavelengths_pdf(l.x), visible_wavelengths_pdf(l.y), visible_wavelengths_pdf(l.z), visible_wavelengths_pdf(l.w)); }
// One 8x8 pixel tile = a contiguous run of 64 * S path slots, by one wave.  The run is read 64 slots at a time (coalesced; the colour
// conversion — twelve IEEE divisions and table reads per sample, the bulk of this kernel — runs on all lanes), the weighted
// colours go through LDS (`mine`: 256 floats of the wave), and the entries of a pixel are added one after the other IN SAMPLE ORDER, as
// the reference adds them (so the film does not depend on the pass size): lanes 4p .. 4p+3 own the four channels of the p-th pixel met
// in the step.
template <typename ACC>
__device__ __forceinline__ void film_tile(const DPathState& st, const DFrame& fr, const DTables& T, ACC* __restrict__ accum, float* __restrict__ mine, int tile) {
    const int lane = lane_id();
    const int S = fr.samples_in_pass;
    const size_t N = (size_t)fr.width * fr.height;
        const size_t base = (size_t)tile * 64 * S;
        // pass sizes that tile the 64-slot steps (S a multiple of 64, or 4 / 8 / 16 / 32): whole pixels per step, one quad per pixel;
        // any other S: one lane per pixel walks the steps.  Both add in sample order.
        const int ch = lane & 3, quad = lane >> 2;
        if (S >= 64 ? (S & 63) == 0 : (S >= 4 && (64 % S) == 0)) {
            const int pix_per_step = S >= 64 ? 1 : 64 / S;        // pixels wholly inside one step (S < 64), or one pixel over S / 64 steps
            const int steps_per_pix = S >= 64 ? S / 64 : 1;
            for (int p0 = 0; p0 < 64; p0 += pix_per_step) {       // first pixel of the group handled together
                ACC acc = 0;
                bool inside = false;
                size_t fp = 0;
                if (quad < pix_per_step) {
                    int px, py;
                    slot_to_pixel(fr, tile * 64 + p0 + quad, px, py, inside);
                    fp = inside ? (size_t)py * fr.width + px : 0;
                    if (inside) acc = ch < 3 ? accum[3 * fp + ch] : accum[3 * N + fp];
                }
                for (int stp = 0; stp < steps_per_pix; ++stp) {
                    const size_t slot = base + (size_t)p0 * S + (size_t)stp * 64 + lane;
                    // slots of film padding were never written by k_camera: whatever they hold is converted but never added
                    const S4 lam_ = ld4(&st.lambda_s[slot]);
                    const v3 rgb = spectral_to_rgb_clamped(T, ld4(&st.L[slot]), lam_, pdf_of(lam_), fr.max_component_value);
                    const float fw = st.filter_w[slot];
                    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                    mine[4 * lane + 0] = fw * rgb.x;
                    mine[4 * lane + 1] = fw * rgb.y;
                    mine[4 * lane + 2] = fw * rgb.z;
                    mine[4 * lane + 3] = fw;
                    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                    if (quad < pix_per_step && inside) {
                        const int first = S >= 64 ? 0 : quad * S, cnt = S >= 64 ? 64 : S;
                        for (int t = 0; t < cnt; ++t) acc += (ACC)mine[4 * (first + t) + ch];
                    }
                }
                if (quad < pix_per_step && inside) {
                    if (ch < 3)
                        accum[3 * fp + ch] = acc;
                    else
                        accum[3 * N + fp] = acc;
                }
            }
        } else {
            // general S: lane = pixel, every 64-slot step's entries of a pixel added by its lane (four channels in turn)
            int px, py;
            bool inside;
            slot_to_pixel(fr, tile * 64 + lane, px, py, inside);
            const size_t p = inside ? (size_t)py * fr.width + px : 0;
            ACC r = 0, g = 0, b = 0, w = 0;
            if (inside) r = accum[3 * p], g = accum[3 * p + 1], b = accum[3 * p + 2], w = accum[3 * N + p];
            const int lo = lane * S, hi = lo + S;
            for (int j = 0; j < 64 * S; j += 64) {
                const size_t slot = base + j + lane;
                const S4 lam_ = ld4(&st.lambda_s[slot]);
                const v3 rgb = spectral_to_rgb_clamped(T, ld4(&st.L[slot]), lam_, pdf_of(lam_), fr.max_component_value);
                const float fw = st.filter_w[slot];
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                mine[4 * lane + 0] = fw * rgb.x;
                mine[4 * lane + 1] = fw * rgb.y;
                mine[4 * lane + 2] = fw * rgb.z;
                mine[4 * lane + 3] = fw;
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                const int a = lo > j ? lo : j, e = hi < j + 64 ? hi : j + 64;
                if (inside)
                    for (int t = a; t < e; ++t) {
                        r += (ACC)mine[4 * (t - j) + 0];
                        g += (ACC)mine[4 * (t - j) + 1];
                        b += (ACC)mine[4 * (t - j) + 2];
                        w += (ACC)mine[4 * (t - j) + 3];
                    }
            }
            if (inside) {
                accum[3 * p] = r;
                accum[3 * p + 1] = g;
                accum[3 * p + 2] = b;
                accum[3 * N + p] = w;
            }
        }
}
template <typename ACC>
__global__ void __launch_bounds__(256) k_film(DPathState st, DFrame fr, DTables T, ACC* __restrict__ accum) {
    __shared__ float buf[4][64 * 4];
    float* mine = buf[threadIdx.x >> 6];
    const int n_tiles = fr.n_pixels_padded >> 6;
    const int n_waves = (int)(gridDim.x * (blockDim.x >> 6));
    for (int tile = (int)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)); tile < n_tiles; tile += n_waves) film_tile<ACC>(st, fr, T, accum, mine, tile);
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
// postprocess_kernel! (src/postprocess.jl:185-250): exposure, white balance, imaging ratio, tone curve, gamma, escaped-ray mask.
// src/dst: Julia [h,w] RGB layout (3 floats per pixel, linear index i = row + h*col); depth likewise.
// ---------------------------------------------------------------------------------------------------
HKD float pp_unch2(float x) {
    const float A = 0.15f, B = 0.50f, C = 0.10f, D = 0.20f, E = 0.02f, F = 0.30f;
    return ((x * (A * x + C * B) + D * E) / (x * (A * x + B) + D * F)) - E / F;
}
HKD float pp_filmic(float x) {
    x = maxf(0.0f, x - 0.004f);
    return (x * (6.2f * x + 0.5f)) / (x * (6.2f * x + 1.7f) + 0.06f);
}
__global__ void __launch_bounds__(256) k_postprocess(hk_postprocess_params P, const float* __restrict__ src, const float* __restrict__ depth, float* __restrict__ dst, int h, int w) {
    const long n = (long)h * w;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        float r = src[3 * i] * P.exposure, g = src[3 * i + 1] * P.exposure, b = src[3 * i + 2] * P.exposure;
        if (P.apply_wb) {
            float ro = P.wb[0] * r + P.wb[1] * g + P.wb[2] * b, go = P.wb[3] * r + P.wb[4] * g + P.wb[5] * b, bo = P.wb[6] * r + P.wb[7] * g + P.wb[8] * b;
            r = maxf(0.0f, ro), g = maxf(0.0f, go), b = maxf(0.0f, bo);
        }
        r = r * P.imaging_ratio, g = g * P.imaging_ratio, b = b * P.imaging_ratio;
        switch (P.tonemap) {
            case HK_TONEMAP_REINHARD: {
                float lum = 0.2126f * r + 0.7152f * g + 0.0722f * b;
                float sc = lum > 0.0f ? 1.0f / (1.0f + lum) : 1.0f;
                r = clampf(r * sc, 0.0f, 1.0f), g = clampf(g * sc, 0.0f, 1.0f), b = clampf(b * sc, 0.0f, 1.0f);
            } break;
            case HK_TONEMAP_REINHARD_EXT: {
                float lum = 0.2126f * r + 0.7152f * g + 0.0722f * b;
                float lw2 = P.white_point * P.white_point;
                float sc = lum > 0.0f ? (1.0f + lum / lw2) / (1.0f + lum) : 1.0f;
                r = clampf(r * sc, 0.0f, 1.0f), g = clampf(g * sc, 0.0f, 1.0f), b = clampf(b * sc, 0.0f, 1.0f);
            } break;
            case HK_TONEMAP_ACES: {
                const float a = 2.51f, bc = 0.03f, c = 2.43f, d = 0.59f, e = 0.14f;
                r = clampf((r * (a * r + bc)) / (r * (c * r + d) + e), 0.0f, 1.0f);
                g = clampf((g * (a * g + bc)) / (g * (c * g + d) + e), 0.0f, 1.0f);
                b = clampf((b * (a * b + bc)) / (b * (c * b + d) + e), 0.0f, 1.0f);
            } break;
            case HK_TONEMAP_UNCHARTED2: {
                float ws = 1.0f / pp_unch2(11.2f);
                r = clampf(pp_unch2(r * 2.0f) * ws, 0.0f, 1.0f), g = clampf(pp_unch2(g * 2.0f) * ws, 0.0f, 1.0f), b = clampf(pp_unch2(b * 2.0f) * ws, 0.0f, 1.0f);
            } break;
            case HK_TONEMAP_FILMIC: r = pp_filmic(r), g = pp_filmic(g), b = pp_filmic(b); break;
            default: r = clampf(r, 0.0f, 1.0f), g = clampf(g, 0.0f, 1.0f), b = clampf(b, 0.0f, 1.0f); break;
        }
        if (P.apply_gamma) r = powf(r, P.inv_gamma), g = powf(g, P.inv_gamma), b = powf(b, P.inv_gamma);
        if (P.mask_escaped && depth) {
            int row = (int)(i % h) + 1, col = (int)(i / h) + 1;
            int d_row = h - row + 1;  // Y flip
            int escaped = 0, total = 0;
            for (int dr = -1; dr <= 1; ++dr)
                for (int dc = -1; dc <= 1; ++dc) {
                    int nr = d_row + dr, nc = col + dc;
                    if (nr >= 1 && nr <= h && nc >= 1 && nc <= w) {
                        escaped += isinf(depth[(long)(nc - 1) * h + nr - 1]) ? 1 : 0;
                        total += 1;
                    }
                }
            float alpha = (float)escaped / (float)total;
            r = r * (1.0f - alpha) + P.bg[0] * alpha, g = g * (1.0f - alpha) + P.bg[1] * alpha, b = b * (1.0f - alpha) + P.bg[2] * alpha;
        }
        dst[3 * i] = r, dst[3 * i + 1] = g, dst[3 * i + 2] = b;
    }
}

// aux_buffer_kernel! (src/film.jl:435-483): first-hit albedo / normal / depth per pixel centre, Julia [h,w] layout
__global__ void __launch_bounds__(HK_TRACE_BLOCK) k_aux(DScene sc, DCamera cam, int h, int w, float miss_depth, float* __restrict__ albedo, float* __restrict__ normal,
                                                      float* __restrict__ depth) {
    __shared__ int lds_stack[(HK_TRACE_BLOCK / 64) * HK_LDS_STACK * 64];
    int* stack = lds_stack + (threadIdx.x >> 6) * (HK_LDS_STACK * 64);
    const int lane = lane_id();
    unsigned a = 0, b = 0;
    const long n = (long)h * w;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        int row = (int)(i % h) + 1, col = (int)(i / h) + 1;
        v2 pixel = mk2(((float)col - 1.0f) + 0.5f, ((float)row - 1.0f) + 0.5f);
        v3 ro, rd;
        float time;
        generate_ray(cam, pixel, mk2(0.5f, 0.5f), 0.0f, ro, rd, time);
        bool opaque;
        HitRec hr = traverse<0, false>(sc, ro, rd, INF_F, stack, lane, a, b, opaque);
        float alb = 0.0f, d = miss_depth;
        v3 nn = mk3(0, 0, 0);
        if (hr.prim >= 0) {
            nn = geometric_normal(sc, hr.prim);
            v3 hp = ro + rd * hr.t;
            v3 dd = hp - ro;
            d = sqrtf(dd.x * dd.x + dd.y * dd.y + dd.z * dd.z);
            alb = 0.8f;
        }
        albedo[3 * i] = albedo[3 * i + 1] = albedo[3 * i + 2] = alb;
        normal[3 * i] = nn.x, normal[3 * i + 1] = nn.y, normal[3 * i + 2] = nn.z;
        depth[i] = d;
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

// resolve_mix_material (mix-material.jl:222-238) for n hit points: out = index of the material a MixMaterial resolves to
__global__ void k_test_mix(DScene sc, int mat_idx, int n, const float* p3, const float* wo3, const float* uv2, int* out) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        out[i] = resolve_mix_material(sc, mat_idx, mk3(p3[3 * i], p3[3 * i + 1], p3[3 * i + 2]), mk3(wo3[3 * i], wo3[3 * i + 1], wo3[3 * i + 2]), mk2(uv2[2 * i], uv2[2 * i + 1]));
}
// media: mode 0 = sample_point (media.jl:1327-1370, 1527-1575; nanovdb.jl:400-469) -> out[13] = sigma_a4, sigma_s4, Le4, g;
//        mode 1 = majorant iterator along a ray (media.jl:229-340, 625-729) -> out[1 + 3*HK_TEST_MAJ_SEGS] = segment count, then
//                 (t_min, t_max, sigma_maj[0]) of the first HK_TEST_MAJ_SEGS segments.  Same MM instantiation as the tracking kernels.
//        mode 2 = the same walk with majorant_skip_zero in front of every majorant_next (what the tracking kernels do): total
//                 segment count incl. the skipped ones, then the first HK_TEST_MAJ_SEGS segments that were NOT skipped.
#define HK_TEST_MAJ_SEGS 16
template <int MM>
__global__ void k_test_medium(DScene sc, DTables T, int mode, int medium_idx, int n, const float* a3, const float* b3, const float* tmax, const float* lambda, float* out) {
    const DMedium& med = sc.media[medium_idx];
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const S4 l = s4(lambda[4 * i], lambda[4 * i + 1], lambda[4 * i + 2], lambda[4 * i + 3]);
        const v3 a = mk3(a3[3 * i], a3[3 * i + 1], a3[3 * i + 2]);
        const S4 base_a = eval_scaled(med.sigma_a, l), base_s = eval_scaled(med.sigma_s, l), base_Le = eval_scaled(med.Le, l);
        if (mode == 0) {
            MediumProps mp = sample_point<MM>(T, l, med, base_a, base_s, base_Le, a);
            float* r = out + 13 * (size_t)i;
            r[0] = mp.sigma_a.x, r[1] = mp.sigma_a.y, r[2] = mp.sigma_a.z, r[3] = mp.sigma_a.w;
            r[4] = mp.sigma_s.x, r[5] = mp.sigma_s.y, r[6] = mp.sigma_s.z, r[7] = mp.sigma_s.w;
            r[8] = mp.Le.x, r[9] = mp.Le.y, r[10] = mp.Le.z, r[11] = mp.Le.w;
            r[12] = mp.g;
        } else {
            const v3 d = mk3(b3[3 * i], b3[3 * i + 1], b3[3 * i + 2]);
            float* r = out + (1 + 3 * HK_TEST_MAJ_SEGS) * (size_t)i;
            for (int k = 0; k < 1 + 3 * HK_TEST_MAJ_SEGS; ++k) r[k] = 0.0f;
            MajorantIter it = create_majorant_iterator<MM>(med, a, d, tmax[i]);
            int count = 0, kept = 0;
            float t0, t1;
            S4 sm;
            for (;;) {
                if (mode == 2) majorant_skip_zero<MM>(it, med, count);   // fast-forward over zero cells (measured and not used by the kernels: DESIGN §5)
                if (count >= 256 || !majorant_next<MM>(it, med, base_a + base_s, t0, t1, sm)) break;
                // mode 1 records every segment, mode 2 the segments that survive the fast-forward (zero cells excluded)
                if (kept < HK_TEST_MAJ_SEGS) r[1 + 3 * kept] = t0, r[2 + 3 * kept] = t1, r[3 + 3 * kept] = sm.x;
                ++kept;
                ++count;
            }
            r[0] = (float)count;
        }
    }
}
// The traversal of the surfaces-only bench path: lane_ray_round (while-while rounds, straggler exit, LDS stack of STACK entries)
// driven by the same per-lane refill as k_trace_lean / k_shadow, over a plain ray array.  Every wave owns a contiguous range of
// rays.  ANYHIT = the shadow kernel's first-accepted-hit mode (out_prim >= 0 <=> occluded).
template <bool ANYHIT, int STACK, bool QN = false>
__global__ void __launch_bounds__(HK_TRACE_BLOCK) k_test_trace_lean(DScene sc, int n, const float* o3, const float* d3, const float* tmax, float* out_t, int* out_prim, float* out_uv) {
    __shared__ int lds_stack[(HK_TRACE_BLOCK / 64) * STACK * 64];
    int* stack = lds_stack + (threadIdx.x >> 6) * (STACK * 64);
    const int lane = lane_id();
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    const int DONE = (int)0x80000000;
    const int waves = physical_waves();
    const int per_wave = (n + waves - 1) / waves;
    const int first = global_wave() * per_wave;
    const int count = first >= n ? 0 : (n - first < per_wave ? n - first : per_wave);
    unsigned n_nodes = 0, n_tris = 0;
    int cursor = 0;
    bool have = false;
    int idx = 0;
    LaneRay r;
    r.cur = r.pend = DONE;
    for (;;) {
        const unsigned long long run_m = __ballot(have && r.cur != DONE);
        if (run_m == 0ull || (64 - __popcll(run_m) >= HK_TRACE_MIN_IDLE && cursor < count)) {
            if (have && r.cur == DONE) {
                out_t[idx] = r.best.prim >= 0 ? r.best.t : INF_F;
                out_prim[idx] = r.best.prim;
                out_uv[2 * idx] = r.best.prim >= 0 ? r.best.u : 0.0f;
                out_uv[2 * idx + 1] = r.best.prim >= 0 ? r.best.v : 0.0f;
                have = false;
            }
            const unsigned long long want = __ballot(!have);
            const int avail = count - cursor;
            const int rank = __popcll(want & lt_mask);
            if (!have && rank < avail) {
                idx = first + cursor + rank;
                lane_ray_start<QN>(r, sc, mk3(o3[3 * idx], o3[3 * idx + 1], o3[3 * idx + 2]), mk3(d3[3 * idx], d3[3 * idx + 1], d3[3 * idx + 2]), tmax[idx]);
                have = true;
            }
            const int want_n = __popcll(want);
            cursor += want_n < avail ? want_n : (avail > 0 ? avail : 0);
            if (__ballot(have) == 0ull) break;
        }
        lane_ray_round<ANYHIT, false, 0, (ANYHIT ? HK_POSTPONE_ANYHIT != 0 : HK_POSTPONE_CLOSEST != 0), false, QN>(r, have && r.cur != DONE, sc, stack, lane, n_nodes, n_tris);
    }
}

// ---------------------------------------------------------------------------------------------------
// denoise! (src/denoise.jl): 3x3 luminance variance (:236-286) and one a-trous pass (:136-229).  Buffers are Julia [h,w]
// column-major: linear index i = (col-1)*h + (row-1), exactly the reference's idx -> (row, col) mapping.
// ---------------------------------------------------------------------------------------------------
HKD float denoise_luminance(float r, float g, float b) { return 0.2126f * r + 0.7152f * g + 0.0722f * b; }
__global__ void __launch_bounds__(256) k_denoise_variance(const float* __restrict__ src, float* __restrict__ variance, int h, int w) {
    const long n = (long)h * w;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        int row = (int)(i % h), col = (int)(i / h);
        float sum = 0.0f, sum_sq = 0.0f;
        int count = 0;
        for (int dy = -1; dy <= 1; ++dy)
            for (int dx = -1; dx <= 1; ++dx) {
                int qr = row + dy, qc = col + dx;
                if (qr >= 0 && qr < h && qc >= 0 && qc < w) {
                    const float* q = src + 3 * ((long)qc * h + qr);
                    float lum = denoise_luminance(q[0], q[1], q[2]);
                    sum += lum;
                    sum_sq += lum * lum;
                    ++count;
                }
            }
        float mean = sum / (float)count, mean_sq = sum_sq / (float)count;
        variance[i] = maxf(0.0f, mean_sq - mean * mean);
    }
}
__global__ void __launch_bounds__(256) k_denoise_atrous(hk_denoise_params P, int step, const float* __restrict__ src, const float* __restrict__ normal,
                                                        const float* __restrict__ depth, const float* __restrict__ variance, float* __restrict__ dst, int h, int w) {
    const float K1D[5] = {1.0f / 16.0f, 1.0f / 4.0f, 3.0f / 8.0f, 1.0f / 4.0f, 1.0f / 16.0f};
    const long n = (long)h * w;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        int row = (int)(i % h), col = (int)(i / h);
        float r_p = src[3 * i], g_p = src[3 * i + 1], b_p = src[3 * i + 2];
        float lum_p = denoise_luminance(r_p, g_p, b_p);
        float nx = normal[3 * i], ny = normal[3 * i + 1], nz = normal[3 * i + 2];
        float d_p = depth[i];
        float var_p = P.use_variance ? variance[i] : 0.0f;
        // weight_color's sigma (:76-88) depends on the centre pixel only
        float sigma_c = var_p > 0.0f ? P.sigma_color * sqrtf(var_p) + 1.0e-4f : P.sigma_color;
        float sigma_d = P.sigma_depth * (float)step + 1.0e-4f;
        float sr = 0.0f, sg = 0.0f, sb = 0.0f, sw = 0.0f;
        for (int dyi = 0; dyi < 5; ++dyi)
            for (int dxi = 0; dxi < 5; ++dxi) {
                int qr = row + (dyi - 2) * step, qc = col + (dxi - 2) * step;
                qr = qr < 0 ? 0 : (qr > h - 1 ? h - 1 : qr);
                qc = qc < 0 ? 0 : (qc > w - 1 ? w - 1 : qc);
                long q = (long)qc * h + qr;
                float r_q = src[3 * q], g_q = src[3 * q + 1], b_q = src[3 * q + 2];
                float lum_q = denoise_luminance(r_q, g_q, b_q);
                float w_spatial = K1D[dxi] * K1D[dyi];
                float w_color = expf(-fabsf(lum_p - lum_q) / sigma_c);
                float dotv = nx * normal[3 * q] + ny * normal[3 * q + 1] + nz * normal[3 * q + 2];
                float w_norm = powf(maxf(0.0f, dotv), P.sigma_normal);
                float w_depth = expf(-fabsf(d_p - depth[q]) / sigma_d);
                float weight = w_spatial * w_color * w_norm * w_depth;
                sr += r_q * weight;
                sg += g_q * weight;
                sb += b_q * weight;
                sw += weight;
            }
        if (sw > 1.0e-6f) {   // false for NaN (centre depth +Inf against +Inf neighbours): the pixel is kept
            float inv = 1.0f / sw;
            dst[3 * i] = sr * inv, dst[3 * i + 1] = sg * inv, dst[3 * i + 2] = sb * inv;
        } else
            dst[3 * i] = r_p, dst[3 * i + 1] = g_p, dst[3 * i + 2] = b_p;
    }
}

// ---------------------------------------------------------------------------------------------------
// launch wrappers (called from hk_api.cpp)
// ---------------------------------------------------------------------------------------------------
namespace hk {

#ifndef HK_OCC_SCALE_DEFAULT
#define HK_OCC_SCALE_DEFAULT 1.0f
#endif
static inline int grid_for(int n, int block, int cap) {
    long g = ((long)n + block - 1) / block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (int)g;
}

// blocks per CU that are actually resident for a kernel (occupancy API, capped): the wave-segment loop makes any
// grid size correct, so the grid is sized to residency instead of oversubscribing and paying a tail round.
template <class K>
static int resident_blocks(K kernel, int block) {   // blocks per CU as the occupancy API reports them (uncapped)
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, block, 0) != hipSuccess || per_cu < 1) per_cu = 1;
    return per_cu;
}
// HK_OCC_SCALE (round 6): the occupancy API of this ROCm answers HALF of what the register file admits for the 256-thread kernels here
// (k_camera, 52 VGPRs: 4 blocks per CU where 512 / 56 registers give 8 waves per SIMD = 8 blocks; MI355X_MICROARCH.md "Register files").
// Grids are sized by `answer x scale`, capped at the caller's cap (8 blocks = 32 waves per CU, the hardware's own limit) — a block the
// CU cannot hold simply starts later, and the ticketed / strided segment loops make any grid size correct.
static float occ_scale() {
    const char* e = hk::knob("HK_OCC_SCALE");
    const float v = e ? (float)std::atof(e) : HK_OCC_SCALE_DEFAULT;
    return v >= 0.25f && v <= 8.0f ? v : 1.0f;
}
// Residency is a property of (kernel, device): cached per kernel instantiation AND per device, so one process can drive several
// GPUs (the in-library multi-device path) without one device's answer leaking to another.
#define HK_MAX_DEVICES 64
template <auto Kernel>
static int cached_blocks(int block, int n_cu, int cap_per_cu) {
    static int cache[HK_MAX_DEVICES];
    int dev = 0, per_cu;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= HK_MAX_DEVICES)
        per_cu = resident_blocks(Kernel, block);
    else {
        if (cache[dev] == 0) {
            cache[dev] = resident_blocks(Kernel, block);
            if (hk::knob("HK_DEBUG_ALLOC")) std::fprintf(stderr, "HK_DEBUG_ALLOC occupancy API: %d blocks of %d per CU for %s\n", cache[dev], block, __PRETTY_FUNCTION__);
        }
        per_cu = cache[dev];
    }
    per_cu = (int)((float)per_cu * occ_scale() + 0.5f);
    if (per_cu < 1) per_cu = 1;
    if (per_cu > cap_per_cu) per_cu = cap_per_cu;
    return per_cu * n_cu;
}
// Segments are walked with a static stride, so the number of physical waves must DIVIDE W or the last round runs with a
// fraction of the waves (shade at W = 16/CU with 12 resident: 12 % slower than at W = 24): the largest divisor of the
// 4-wave block count that is resident.
static int node_cache_mode() {   // HK_NODE_CACHE=0: the lean closest-hit kernel reads every node from global memory (A/B switch, looked up per launch in the context's knob table)
    const char* e = hk::knob("HK_NODE_CACHE");
    return (e && std::atoi(e) == 0) ? 0 : 1;
}
static int clamp_blocks(int blocks, const DPathState& st, int waves_per_block = 4) {
    const int units = st.n_waves / waves_per_block;
    if (blocks >= units) return units;
    if (st.dynamic_segments) return blocks;
    int best = 1;
    for (int b = blocks; b >= 1; --b)
        if (units % b == 0) {
            best = b;
            break;
        }
    // a divisor far below residency wastes more than the tail round it avoids
    return best * 4 >= blocks * 3 ? best : blocks;
}

void launch_camera(hipStream_t s, int n_cu, const DPathState& st, const DFrame& fr, const DTables& T, const DFilter& f, const DCamera& c, const DSobol& sob, int initial_medium) {
    const int blocks = cached_blocks<k_camera>(256, n_cu, 8);
    hipLaunchKernelGGL(k_camera, dim3(clamp_blocks(blocks, st)), dim3(256), 0, s, st, fr, T, f, c, sob, initial_medium);
}
void launch_trace(hipStream_t s, int n_cu, const DPathState& st, const DScene& sc, const DTables& T, const DFrame& fr, int depth, DStats* stats) {
    if (sc.all_opaque && sc.n_media == 0) {
#define HK_LEAN_LAUNCH(K, C, S, B, NC)                                                                                         \
    {                                                                                                                          \
        const int blocks = cached_blocks<K<C, S, B, NC>>(B, n_cu, 8);                                                          \
        hipLaunchKernelGGL((K<C, S, B, NC>), dim3(clamp_blocks(blocks, st, B / 64)), dim3(B), 0, s, st, sc, depth, stats);     \
    }
#define HK_LEAN_LAUNCH_QN(K, C, S, B, NC)   /* the quantised nodes of a deep tree (DScene::qnodes) */                            \
    {                                                                                                                          \
        const int blocks = cached_blocks<K<C, S, B, NC, true>>(B, n_cu, 8);                                                    \
        hipLaunchKernelGGL((K<C, S, B, NC, true>), dim3(clamp_blocks(blocks, st, B / 64)), dim3(B), 0, s, st, sc, depth, stats); \
    }
// k_trace_lean over a BVH of depth <= 16: ONE 1024-thread block per CU (the 4 waves per SIMD its registers allow anyway) whose LDS
// holds the 16 stacks (64 KB) and the top 1536 nodes of the tree (84 KB) — all of the Cornell box, the upper levels of the others.
// Measured (HK_NODE_CACHE=0 switches it off): trace -11 % in the Cornell box and the sky scene; with the 32-entry stacks of the
// 10^6-triangle scene 512 nodes are a small part of the visits (+-1 %: no cache there).
// The any-hit kernel lives on residency (7 waves per SIMD on 16 KB of stacks; the big cache cost it three of them: +3 %), so it
// gets the cache that FITS beside them.  Round 6 (two-spheres Cornell box, A B | B A): what counts is nodes per WAVE-SLOT — a cache is shared by the
// waves of its block, so bigger blocks buy more of the tree for the same LDS.  12-wave blocks with the top 560 nodes (48 KB of stacks + 31 KB of
// nodes, two blocks per CU = 6 waves per SIMD): k_shadow 17.55 -> 16.4 ms per Cornell frame, sky 2.6 -> 2.5.  On the way: 4-wave blocks with
// 112 nodes (rounds 3-5, seven blocks per CU) 17.55, 176 nodes (six blocks) 17.0, 192 / 240 / 320 nodes (five, five, four blocks) 18.5 / 18.3 /
// 20.3; 8-wave blocks with 384 nodes 20.2 and 14-wave blocks with 432 nodes 22.5 (one block short of the plan each); 16-wave blocks with the
// whole tree (four waves per SIMD) 17.0.
#ifndef HK_SHADOW_NC16
#define HK_SHADOW_NC16 560
#endif
#ifndef HK_SHADOW_BLOCK16   // threads per block of the any-hit kernel over trees <= 16 deep
#define HK_SHADOW_BLOCK16 768
#endif
#ifndef HK_SHADOW_NC32
#define HK_SHADOW_NC32 0
#endif
#ifndef HK_TRACE_NC32
#define HK_TRACE_NC32 128   // 10^6-triangle scene: trace -1.3 % (256 nodes cost a block per CU: +15 %)
#endif
#define HK_LEAN_DISPATCH(K, B16, NC16, NC32)                                     \
    if (sc.bvh_depth <= 16) {                                                    \
        if (node_cache_mode() != 0) {                                            \
            if (fr.count_nodes) HK_LEAN_LAUNCH(K, true, 16, B16, NC16)           \
            else HK_LEAN_LAUNCH(K, false, 16, B16, NC16)                         \
        } else {                                                                 \
            if (fr.count_nodes) HK_LEAN_LAUNCH(K, true, 16, HK_TRACE_BLOCK, 0)   \
            else HK_LEAN_LAUNCH(K, false, 16, HK_TRACE_BLOCK, 0)                 \
        }                                                                        \
    } else if (sc.qnodes != nullptr) {                                           \
        if (node_cache_mode() != 0) {                                            \
            if (fr.count_nodes) HK_LEAN_LAUNCH_QN(K, true, HK_LDS_STACK, HK_TRACE_BLOCK, NC32)    \
            else HK_LEAN_LAUNCH_QN(K, false, HK_LDS_STACK, HK_TRACE_BLOCK, NC32) \
        } else {   /* HK_NODE_CACHE=0 is honoured for the quantised tree too */ \
            if (fr.count_nodes) HK_LEAN_LAUNCH_QN(K, true, HK_LDS_STACK, HK_TRACE_BLOCK, 0)   \
            else HK_LEAN_LAUNCH_QN(K, false, HK_LDS_STACK, HK_TRACE_BLOCK, 0)    \
        }                                                                        \
    } else {                                                                     \
        if (node_cache_mode() != 0) {                                            \
            if (fr.count_nodes) HK_LEAN_LAUNCH(K, true, HK_LDS_STACK, HK_TRACE_BLOCK, NC32)   \
            else HK_LEAN_LAUNCH(K, false, HK_LDS_STACK, HK_TRACE_BLOCK, NC32)    \
        } else {                                                                 \
            if (fr.count_nodes) HK_LEAN_LAUNCH(K, true, HK_LDS_STACK, HK_TRACE_BLOCK, 0)   \
            else HK_LEAN_LAUNCH(K, false, HK_LDS_STACK, HK_TRACE_BLOCK, 0)       \
        }                                                                        \
    }
        HK_LEAN_DISPATCH(k_trace_lean, 1024, 1536, HK_TRACE_NC32)
        return;
    }
    const int b0 = cached_blocks<k_trace<false>>(HK_TRACE_BLOCK, n_cu, 8), b1 = cached_blocks<k_trace<true>>(HK_TRACE_BLOCK, n_cu, 8);
    if (fr.count_nodes)
        hipLaunchKernelGGL(k_trace<true>, dim3(clamp_blocks(b1, st)), dim3(HK_TRACE_BLOCK), 0, s, st, sc, T, fr, depth, stats);
    else
        hipLaunchKernelGGL(k_trace<false>, dim3(clamp_blocks(b0, st)), dim3(HK_TRACE_BLOCK), 0, s, st, sc, T, fr, depth, stats);
}
static int grey_mode() {   // HK_GREY=0: flat-spectrum media run through the general tracking kernels (A/B switch, looked up per launch in the context's knob table)
    const char* e = hk::knob("HK_GREY");
    return e ? std::atoi(e) : 1;
}
static int grey_flat_mode() {   // HK_GREY_FLAT=0: grey media run through the round-3 GREY instantiations of k_track / k_shadow_walk (A/B switch, looked up per launch in the context's knob table)
    const char* e = hk::knob("HK_GREY_FLAT");
    return e ? std::atoi(e) : 1;
}
static int walk_pool_mode() {   // HK_WALK_POOL=0: k_shadow_walk<.., GREY> instead of k_walk_pool (A/B switch, looked up per launch in the context's knob table)
    const char* e = hk::knob("HK_WALK_POOL");
    return e ? std::atoi(e) : 1;
}
static int track_pool_mode() {   // HK_TRACK_POOL=0: k_track_flat instead of k_track_pool (A/B switch, looked up per launch in the context's knob table)
    const char* e = hk::knob("HK_TRACK_POOL");
    return e ? std::atoi(e) : 1;
}
// media kernels are instantiated for a single medium kind or for all four (15)
static int media_mask_class(const DScene& sc) {
    int m = sc.media_mask;
    return (m == 1 || m == 2 || m == 4 || m == 8) ? m : 15;
}
static int walk_split_mode() {   // HK_WALK_SPLIT=1: the grey medium's shadow walk runs as k_walk_cast / k_walk_track rounds instead of ONE k_shadow_walk<.., GREY>
    const char* e = hk::knob("HK_WALK_SPLIT");   // (measured on the BOMEX stand-in: cast rounds 0.085 s + tracking rounds 0.45 s against 0.51 s unsplit — off by default)
    return e ? std::atoi(e) : 0;
}
template <bool C, int MM>
static void launch_walk_split(hipStream_t s, int n_cu, const DPathState& st, const DScene& sc, const DFrame& fr, int depth, DStats* stats) {
    const int cb = clamp_blocks(cached_blocks<k_walk_cast<C, MM, 16>>(HK_TRACE_BLOCK, n_cu, 8), st);
    const int tb = clamp_blocks(cached_blocks<k_walk_track<MM>>(256, n_cu, 8), st);
    for (int round = 0; round < HK_WALK_ROUNDS; ++round) {
        hipLaunchKernelGGL((k_walk_cast<C, MM, 16>), dim3(cb), dim3(HK_TRACE_BLOCK), 0, s, st, sc, depth, round, stats, sc.media);
        hipLaunchKernelGGL((k_walk_track<MM>), dim3(tb), dim3(256), 0, s, st, sc, depth, round, fr.walk_tune, stats, sc.media);
    }
}
// The compact records (r_u == 1 not stored, r_l one float: DPathState::compact) also hold in a scene whose one medium is grey AND runs
// through the pool kernels (k_track_pool / k_walk_pool, the only media kernels that know the compact layout): r_u is never touched
// there and every factor of r_l is a scalar.  HK_GREY_COMPACT=0: the full records.
bool grey_compact_ok(const DScene& sc) {
    const char* e = hk::knob("HK_GREY_COMPACT");
    if (e && std::atoi(e) == 0) return false;
    const int mc = sc.n_media > 0 ? media_mask_class(sc) : 0;
    return sc.n_media == 1 && sc.all_grey && sc.grey_pool && (mc == 2 || mc == 8) && sc.bvh_depth <= 16 && grey_mode() && grey_flat_mode() && track_pool_mode() && walk_pool_mode() &&
           !walk_split_mode();
}
void launch_shadow(hipStream_t s, int n_cu, const DPathState& st, const DScene& sc, const DTables& T, const DFrame& fr, int depth, DStats* stats) {
    if (sc.all_opaque && sc.n_media == 0) {
        HK_LEAN_DISPATCH(k_shadow, HK_SHADOW_BLOCK16, HK_SHADOW_NC16, HK_SHADOW_NC32)
        return;
    }
    if (sc.all_grey && grey_mode() && walk_split_mode() && sc.bvh_depth <= 16 && st.wq_a != nullptr) {
        const int mc = media_mask_class(sc);
        if (mc == 8) {
            if (fr.count_nodes) launch_walk_split<true, 8>(s, n_cu, st, sc, fr, depth, stats); else launch_walk_split<false, 8>(s, n_cu, st, sc, fr, depth, stats);
            return;
        }
        if (mc == 2) {
            if (fr.count_nodes) launch_walk_split<true, 2>(s, n_cu, st, sc, fr, depth, stats); else launch_walk_split<false, 2>(s, n_cu, st, sc, fr, depth, stats);
            return;
        }
    }
#define HK_SHADOW_LAUNCH(C, MM)                                                                                                        \
    if (sc.bvh_depth <= 16 && (MM == 2 || MM == 8) && sc.grey_pool && grey_mode() && grey_flat_mode() && walk_pool_mode()) {         \
        constexpr int M2 = (MM == 2 || MM == 8) ? MM : 8;                                                                              \
        if (M2 == 8 && sc.grey_bricks && sc.bvh_depth <= 8 && walk_pool_mode() == 1) {                                                 \
            const int blocks = cached_blocks<k_walk_pool<C, 8, true, 8, 64>>(HK_TRACE_BLOCK, n_cu, 8);                                 \
            hipLaunchKernelGGL((k_walk_pool<C, 8, true, 8, 64>), dim3(clamp_blocks(blocks, st)), dim3(HK_TRACE_BLOCK), 0, s, st, sc, depth, fr.walk_tune, fr.track_gate, stats, sc.media); \
        } else if (M2 == 8 && sc.grey_bricks) {                                                                                        \
            const int blocks = cached_blocks<k_walk_pool<C, 8, true, 16, 56>>(HK_TRACE_BLOCK, n_cu, 8);                                \
            hipLaunchKernelGGL((k_walk_pool<C, 8, true, 16, 56>), dim3(clamp_blocks(blocks, st)), dim3(HK_TRACE_BLOCK), 0, s, st, sc, depth, fr.walk_tune, fr.track_gate, stats, sc.media); \
        } else {                                                                                                                       \
            const int blocks = cached_blocks<k_walk_pool<C, M2, false, 16, 56>>(HK_TRACE_BLOCK, n_cu, 8);                              \
            hipLaunchKernelGGL((k_walk_pool<C, M2, false, 16, 56>), dim3(clamp_blocks(blocks, st)), dim3(HK_TRACE_BLOCK), 0, s, st, sc, depth, fr.walk_tune, fr.track_gate, stats, sc.media); \
        }                                                                                                                              \
    } else if (sc.bvh_depth <= 16 && (MM == 2 || MM == 8) && sc.all_grey && grey_mode()) {                                           \
        constexpr int M2 = (MM == 2 || MM == 8) ? MM : 8;                                                                              \
        const int blocks = cached_blocks<k_shadow_walk<C, M2, 16, true>>(HK_TRACE_BLOCK, n_cu, 8);                                   \
        hipLaunchKernelGGL((k_shadow_walk<C, M2, 16, true>), dim3(clamp_blocks(blocks, st)), dim3(HK_TRACE_BLOCK), 0, s, st, sc, T, depth, fr.walk_tune, stats, sc.media); \
    } else if (sc.bvh_depth <= 16) {   /* 16-entry stacks: 16 KB per block */                                                       \
        const int blocks = cached_blocks<k_shadow_walk<C, MM, 16>>(HK_TRACE_BLOCK, n_cu, 8);                                         \
        hipLaunchKernelGGL((k_shadow_walk<C, MM, 16>), dim3(clamp_blocks(blocks, st)), dim3(HK_TRACE_BLOCK), 0, s, st, sc, T, depth, fr.walk_tune, stats, sc.media); \
    } else {                                                                                                                           \
        const int blocks = cached_blocks<k_shadow_walk<C, MM>>(HK_TRACE_BLOCK, n_cu, 8);                                             \
        hipLaunchKernelGGL((k_shadow_walk<C, MM>), dim3(clamp_blocks(blocks, st)), dim3(HK_TRACE_BLOCK), 0, s, st, sc, T, depth, fr.walk_tune, stats, sc.media); \
    }
#define HK_SHADOW_MM(C)                                   \
    switch (sc.n_media > 0 ? media_mask_class(sc) : 0) {  \
        case 0: HK_SHADOW_LAUNCH(C, 0) break;             \
        case 1: HK_SHADOW_LAUNCH(C, 1) break;             \
        case 2: HK_SHADOW_LAUNCH(C, 2) break;             \
        case 4: HK_SHADOW_LAUNCH(C, 4) break;             \
        case 8: HK_SHADOW_LAUNCH(C, 8) break;             \
        default: HK_SHADOW_LAUNCH(C, 15) break;           \
    }
    if (fr.count_nodes) {
        HK_SHADOW_MM(true)
    } else {
        HK_SHADOW_MM(false)
    }
#undef HK_SHADOW_MM
#undef HK_SHADOW_LAUNCH
}
// work lists of up to HK_MAX_KINDS + 6 queues (pairs depth, queue id) in one launch
void launch_segment_lists(hipStream_t s, const DPathState& st, int n, const int* depths, const int* queues) {
    if (n <= 0) return;
    SegQueues qs;
    qs.n = n;
    for (int i = 0; i < n; ++i) {
        qs.depth[i] = depths[i];
        qs.q[i] = queues[i];
    }
    hipLaunchKernelGGL(k_segment_lists, dim3(n, HK_LIST_SPLIT), dim3(1024), 0, s, st, qs);
}
void launch_medium(hipStream_t s, int n_cu, const DPathState& st, const DScene& sc, const DTables& T, const DFrame& fr, const DSobol& sob, int depth, DStats* stats) {
    {
        const int d = depth, q = Q_MEDIUM;
        launch_segment_lists(s, st, 1, &d, &q);
    }
#define COMMA ,
#define HK_TRACK_LAUNCH(MM)                                                                                              \
    {                                                                                                                    \
        const int blocks = cached_blocks<k_track<MM>>(256, n_cu, 8);                                                   \
        hipLaunchKernelGGL((k_track<MM>), dim3(clamp_blocks(blocks, st)), dim3(256), 0, s, st, sc, T, fr, depth, stats, sc.media);   \
    }
#define HK_TRACK_FLAT(MM, B)                                                                                             \
    if (pool) {                                                                                                          \
        const int blocks = cached_blocks<k_track_pool<MM, B>>(256, n_cu, 8);                                           \
        hipLaunchKernelGGL((k_track_pool<MM, B>), dim3(clamp_blocks(blocks, st)), dim3(256), 0, s, st, sc, fr, depth, stats, sc.media);   \
    } else {                                                                                                             \
        const int blocks = cached_blocks<k_track_flat<MM, B>>(256, n_cu, 8);                                           \
        hipLaunchKernelGGL((k_track_flat<MM, B>), dim3(clamp_blocks(blocks, st)), dim3(256), 0, s, st, sc, fr, depth, stats, sc.media);   \
    }
    const bool grey = sc.all_grey && grey_mode();
    const bool flat = grey && grey_flat_mode();
    const bool pool = flat && track_pool_mode() && sc.grey_pool;
    switch (media_mask_class(sc)) {
        case 1: HK_TRACK_LAUNCH(1) break;
        case 2: if (flat) HK_TRACK_FLAT(2, false) else if (grey) HK_TRACK_LAUNCH(2 COMMA true) else HK_TRACK_LAUNCH(2) break;
        case 4: HK_TRACK_LAUNCH(4) break;
        case 8: if (flat && sc.grey_bricks) HK_TRACK_FLAT(8, true) else if (flat) HK_TRACK_FLAT(8, false) else if (grey) HK_TRACK_LAUNCH(8 COMMA true) else HK_TRACK_LAUNCH(8) break;
        default: HK_TRACK_LAUNCH(15) break;
    }
#undef HK_TRACK_LAUNCH
#undef HK_TRACK_FLAT
    {
        const int d = depth, q = Q_SCATTER;
        launch_segment_lists(s, st, 1, &d, &q);
    }
    const char* ft_env = hk::knob("HK_SOBOL_TABLE_ONLY");   // looked up per launch in the context's knob table (A/B switch)
    if (sob.hi_table != nullptr && sob.lo_table != nullptr && 9 + 5 * depth < sob.lo_rows && 9 + 5 * depth < sob.hi_rows && !(ft_env && std::atoi(ft_env) == 0)) {
        const int sblocks = cached_blocks<k_scatter<true>>(256, n_cu, 8);
        hipLaunchKernelGGL(k_scatter<true>, dim3(clamp_blocks(sblocks, st)), dim3(256), 0, s, st, sc, T, fr, sob, depth, stats);
    } else {
        const int sblocks = cached_blocks<k_scatter<false>>(256, n_cu, 8);
        hipLaunchKernelGGL(k_scatter<false>, dim3(clamp_blocks(sblocks, st)), dim3(256), 0, s, st, sc, T, fr, sob, depth, stats);
    }
}
void launch_detect_camera_medium(hipStream_t s, const DPathState& st, const DScene& sc, float x, float y, float z, DStats* stats) {
    hipLaunchKernelGGL(k_detect_camera_medium, dim3(1), dim3(64), 0, s, st, sc, x, y, z, stats);
}
void launch_escaped(hipStream_t s, int n_cu, const DPathState& st, const DScene& sc, const DTables& T, const DFrame& fr, int depth) {
    const char* e = hk::knob("HK_ESCAPED_UNROLL");   // 2: two queue entries per lane and iteration (A/B switch; films bit-identical).  Measured, sky: 8.9 ->
    if (!(e && std::atoi(e) == 2)) {                  // 10.4 ms per frame — its registers cost the kernel a block per CU (3 instead of 4); off
        const int blocks = cached_blocks<k_escaped<1>>(256, n_cu, 8);
        hipLaunchKernelGGL(k_escaped<1>, dim3(clamp_blocks(blocks, st)), dim3(256), 0, s, st, sc, T, depth, fr.implicit_ones);   // (one wave per segment instead of the resident stride: +-0, sky)
    } else {
        const int blocks = cached_blocks<k_escaped<2>>(256, n_cu, 8);
        hipLaunchKernelGGL(k_escaped<2>, dim3(clamp_blocks(blocks, st)), dim3(256), 0, s, st, sc, T, depth, fr.implicit_ones);
    }
}
static bool sobol_tables_cover(const DSobol& sob, int depth) {
    const char* ft_env = hk::knob("HK_SOBOL_TABLE_ONLY");   // looked up per launch in the context's knob table (A/B switch)
    return sob.hi_table != nullptr && sob.lo_table != nullptr && 9 + 5 * depth < sob.lo_rows && 9 + 5 * depth < sob.hi_rows && !(ft_env && std::atoi(ft_env) == 0);
}
// scenes whose next-event light is chosen by k_light_select before the shade kernels run (HK_PRESELECT=0: never)
bool preselect_lights(const DScene& sc, const DPathState& st) {
    const char* e = hk::knob("HK_PRESELECT");
    return sc.n_media == 0 && sc.num_bvh_lights >= HK_PRESELECT_MIN && st.sel_light != nullptr && !(e && std::atoi(e) == 0);
}
void launch_light_select(hipStream_t s, int n_cu, const DPathState& st, const DScene& sc, const DTables& T, const DFrame& fr, const DSobol& sob, int depth, uint32_t kinds_mask, DStats* stats) {
    int min_idle = HK_SELECT_MIN_IDLE;
    if (const char* e = hk::knob("HK_SELECT_MIN_IDLE")) min_idle = std::atoi(e) >= 1 && std::atoi(e) <= 64 ? std::atoi(e) : min_idle;
    const char* pe = hk::knob("HK_SELECT_POOL");   // 0: the per-lane-refill kernel of round 4 (A/B switch; films bit-identical)
    if (!(pe && std::atoi(pe) == 0)) {
        if (sobol_tables_cover(sob, depth)) {
            const int blocks = cached_blocks<k_light_select_pool<true>>(HK_SELECT_BLOCK, n_cu, 8);
            hipLaunchKernelGGL(k_light_select_pool<true>, dim3(clamp_blocks(blocks, st, HK_SELECT_BLOCK / 64)), dim3(HK_SELECT_BLOCK), 0, s, st, sc, T, fr, sob, depth, kinds_mask, stats);
        } else {
            const int blocks = cached_blocks<k_light_select_pool<false>>(HK_SELECT_BLOCK, n_cu, 8);
            hipLaunchKernelGGL(k_light_select_pool<false>, dim3(clamp_blocks(blocks, st, HK_SELECT_BLOCK / 64)), dim3(HK_SELECT_BLOCK), 0, s, st, sc, T, fr, sob, depth, kinds_mask, stats);
        }
        return;
    }
    if (sobol_tables_cover(sob, depth)) {
        const int blocks = cached_blocks<k_light_select<true>>(256, n_cu, 8);
        hipLaunchKernelGGL(k_light_select<true>, dim3(clamp_blocks(blocks, st)), dim3(256), 0, s, st, sc, T, fr, sob, depth, kinds_mask, min_idle, stats);
    } else {
        const int blocks = cached_blocks<k_light_select<false>>(256, n_cu, 8);
        hipLaunchKernelGGL(k_light_select<false>, dim3(clamp_blocks(blocks, st)), dim3(256), 0, s, st, sc, T, fr, sob, depth, kinds_mask, min_idle, stats);
    }
}
void launch_shade(hipStream_t s, int n_cu, int kind, const DPathState& st, const DScene& sc, const DTables& T, const DFrame& fr, const DSobol& sob, int depth, int first_kind, DStats* stats) {
#define HK_SHADE_CASE(K)                                                                                                          \
    case K: {                                                                                                                     \
        const int blocks = cached_blocks<k_shade<K>>(256, n_cu, 8);                                                            \
        hipLaunchKernelGGL(k_shade<K>, dim3(clamp_blocks(blocks, st)), dim3(256), 0, s, st, sc, T, fr, sob, depth, first_kind, stats); \
    } break;
    // both sampler tables hold every draw of this bounce (rows up to 9 + 5 depth: sobol_row) for every path of the pass: the table-only
    // instantiation (HK_SOBOL_TABLE_ONLY=0: always the general one — A/B switch, looked up per launch in the context's knob table)
    const bool ft = sobol_tables_cover(sob, depth);
    if (preselect_lights(sc, st)) {   // k_light_select has run for this depth: the instantiations without the light-BVH descent
#define HK_SHADE_PRE(K)                                                                                                                 \
    if (kind == K) {                                                                                                                    \
        if (ft) {                                                                                                                       \
            const int blocks = cached_blocks<k_shade<K, false, true, true>>(256, n_cu, 8);                                            \
            hipLaunchKernelGGL((k_shade<K, false, true, true>), dim3(clamp_blocks(blocks, st)), dim3(256), 0, s, st, sc, T, fr, sob, depth, first_kind, stats); \
        } else {                                                                                                                        \
            const int blocks = cached_blocks<k_shade<K, false, false, true>>(256, n_cu, 8);                                           \
            hipLaunchKernelGGL((k_shade<K, false, false, true>), dim3(clamp_blocks(blocks, st)), dim3(256), 0, s, st, sc, T, fr, sob, depth, first_kind, stats); \
        }                                                                                                                               \
        return;                                                                                                                         \
    }
        HK_SHADE_PRE(HK_MAT_MATTE)
        HK_SHADE_PRE(HK_MAT_MIRROR)
        HK_SHADE_PRE(HK_MAT_GLASS)
        HK_SHADE_PRE(HK_MAT_CONDUCTOR)
#undef HK_SHADE_PRE
    }
    if (kind == HK_MAT_MATTE && sc.simple_lights) {
        if (ft) {
            const int blocks = cached_blocks<k_shade<HK_MAT_MATTE, true, true>>(256, n_cu, 8);
            hipLaunchKernelGGL((k_shade<HK_MAT_MATTE, true, true>), dim3(clamp_blocks(blocks, st)), dim3(256), 0, s, st, sc, T, fr, sob, depth, first_kind, stats);
        } else {
            const int blocks = cached_blocks<k_shade<HK_MAT_MATTE, true>>(256, n_cu, 8);
            hipLaunchKernelGGL((k_shade<HK_MAT_MATTE, true>), dim3(clamp_blocks(blocks, st)), dim3(256), 0, s, st, sc, T, fr, sob, depth, first_kind, stats);
        }
        return;
    }
#define HK_SHADE_FT(K)                                                                                                                  \
    if (kind == K && ft) {                                                                                                              \
        const int blocks = cached_blocks<k_shade<K, false, true>>(256, n_cu, 8);                                                      \
        hipLaunchKernelGGL((k_shade<K, false, true>), dim3(clamp_blocks(blocks, st)), dim3(256), 0, s, st, sc, T, fr, sob, depth, first_kind, stats); \
        return;                                                                                                                         \
    }
    HK_SHADE_FT(HK_MAT_MATTE)
    HK_SHADE_FT(HK_MAT_MIRROR)
    HK_SHADE_FT(HK_MAT_GLASS)
    HK_SHADE_FT(HK_MAT_CONDUCTOR)
#undef HK_SHADE_FT
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
            const int blocks = cached_blocks<k_shade<HK_MAT_FALLBACK>>(256, n_cu, 8);
            hipLaunchKernelGGL(k_shade<HK_MAT_FALLBACK>, dim3(clamp_blocks(blocks, st)), dim3(256), 0, s, st, sc, T, fr, sob, depth, first_kind, stats);
        } break;
    }
#undef HK_SHADE_CASE
}
// k_small_pass: the whole pass of a small call in one launch.  -> false when this scene / pass is not its case (the caller launches the stages)
#ifndef HK_SMALL_PASS_NC
#define HK_SMALL_PASS_NC 1536
#endif
// the scenes k_small_pass is instantiated for (the pass itself must be a small one: launch_small_pass): -> 0 none, 1 the all-matte closed
// scene under simple lights, 2 the general instantiation (Matte / Mirror / Glass / Conductor, escape lights)
static int small_pass_class(const DScene& sc, uint32_t kinds_mask) {
    const char* e = hk::knob("HK_SMALL_PASS_FUSED");   // 0: the stages as launches, 1: only the all-matte instantiation (A/B switch; films bit-identical)
    const int mode = e ? std::atoi(e) : 2;
    if (mode == 0) return 0;
    if (!(sc.all_opaque && sc.n_media == 0 && sc.bvh_depth <= HK_LDS_STACK && node_cache_mode() != 0)) return 0;
    if (kinds_mask == (1u << HK_MAT_MATTE) && sc.simple_lights && !sc.has_escape_lights && sc.bvh_depth <= 16 && sc.num_bvh_lights < HK_PRESELECT_MIN) return 1;
    const uint32_t light_kinds = (1u << HK_MAT_MATTE) | (1u << HK_MAT_MIRROR) | (1u << HK_MAT_GLASS) | (1u << HK_MAT_CONDUCTOR);
    return (mode >= 2 && kinds_mask != 0 && (kinds_mask & ~light_kinds) == 0) ? 2 : 0;
}
bool small_pass_fusable(const DScene& sc, uint32_t kinds_mask) { return small_pass_class(sc, kinds_mask) != 0; }
bool launch_small_pass(hipStream_t s, int n_cu, const DPathState& st, const DScene& sc, const DTables& T, const DFrame& fr, const DFilter& flt, const DCamera& cam, const DSobol& sob, int max_depth,
                       uint32_t kinds_mask, DStats* stats, bool dry, void* accum, int film_mode) {   // dry: only say whether this pass would be launched; film_mode: 0 the caller launches k_film, 1 / 2: float / double accumulators, added inside (one-sample passes)
    const int cls = small_pass_class(sc, kinds_mask);
    if (cls == 0 || !st.small_pass || st.dynamic_segments || fr.count_nodes) return false;
    // a SMALL pass in ensure_state's sense (at most 16 chunks for each of 4 waves per CU), not the mid-size pass that shares its static
    // segments: with 25 chunks per stage the launches' higher residency wins (800^2 x 32 spp without sampler tables: 14.8 ms as launches, 17.2 here)
    if (((long)st.capacity + 63) / 64 / 16 > 4L * n_cu) return false;
    for (int depth = 0; depth < max_depth; ++depth)
        if (sobol_tables_cover(sob, depth)) return false;
    // one block per CU (its LDS: 16-entry stacks, the top of the tree, the emission lists) of as many waves as the pass has segments per CU
    const char* me = hk::knob("HK_SMALL_PASS_MERGED");   // 0: shadow rays and the next bounce's rays as two stages (A/B switch; films bit-identical)
    const int merged = (me && std::atoi(me) == 0) ? 0 : 1;
#define HK_SMALL_PASS_LAUNCH(B, G, ...)                                                                                                                     \
    {                                                                                                                                                       \
        int blocks = cached_blocks<k_small_pass<HK_SMALL_PASS_NC, B, G, ##__VA_ARGS__>>(B, n_cu, 1);                                                        \
        if (blocks * (B / 64) > st.n_waves) blocks = st.n_waves / (B / 64);                                                                                 \
        if (blocks < 1) return false;                                                                                                                       \
        if (!dry) hipLaunchKernelGGL((k_small_pass<HK_SMALL_PASS_NC, B, G, ##__VA_ARGS__>), dim3(blocks), dim3(B), 0, s, st, sc, T, fr, flt, cam, sob, max_depth, sc.n_lights > 0 ? 1 : 0, merged, accum, film_mode, kinds_mask, stats); \
    }
    // (Cornell 800^2, one sample per call: 1.42 / 1.05 / 1.14 ms at 4 / 8 / 16 segments per CU — 256- / 512- / 1024-thread blocks; a 768-thread
    // block for 12: 1.55.  Eight waves per CU hold 312 paths each: chunks stay fuller down the bounces than with 156, and at two waves per
    // SIMD nothing spills.)
    // two waves per SIMD at most: at 16 segments per CU (1 024-thread blocks, 128 registers) the matte instantiation spilled 372 B per lane
    // and lost to 8 (1.14 ms against 1.05), the general one does not fit at all — such a pass keeps the launches
    if (st.n_waves > 8 * n_cu) return false;
    if (cls == 2) {
        if (sc.bvh_depth > 16) {
            if (st.n_waves > 4 * n_cu) HK_SMALL_PASS_LAUNCH(512, true, HK_LDS_STACK)
            else HK_SMALL_PASS_LAUNCH(256, true, HK_LDS_STACK)
        } else if (st.n_waves > 4 * n_cu) HK_SMALL_PASS_LAUNCH(512, true)
        else HK_SMALL_PASS_LAUNCH(256, true)
        return true;
    }
    if (st.n_waves > 4 * n_cu) HK_SMALL_PASS_LAUNCH(512, false)
    else HK_SMALL_PASS_LAUNCH(256, false)
#undef HK_SMALL_PASS_LAUNCH
    return true;
}
void launch_film(hipStream_t s, const DPathState& st, const DFrame& fr, const DTables& T, void* accum, bool f64) {
    int g = grid_for(fr.n_pixels_padded >> 6, 4, 4096);   // one wave per 8x8 tile
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
void launch_test_mix(hipStream_t s, const DScene& sc, int mat_idx, int n, const float* p3, const float* wo3, const float* uv2, int* out) {
    hipLaunchKernelGGL(k_test_mix, dim3(grid_for(n, 256, 1024)), dim3(256), 0, s, sc, mat_idx, n, p3, wo3, uv2, out);
}
void launch_test_medium(hipStream_t s, const DScene& sc, const DTables& T, int mode, int medium_idx, int n, const float* a3, const float* b3, const float* tmax, const float* lambda,
                        float* out) {
#define HK_TM_LAUNCH(MM) hipLaunchKernelGGL((k_test_medium<MM>), dim3(grid_for(n, 64, 4096)), dim3(64), 0, s, sc, T, mode, medium_idx, n, a3, b3, tmax, lambda, out)
    switch (media_mask_class(sc)) {
        case 1: HK_TM_LAUNCH(1); break;
        case 2: HK_TM_LAUNCH(2); break;
        case 4: HK_TM_LAUNCH(4); break;
        case 8: HK_TM_LAUNCH(8); break;
        default: HK_TM_LAUNCH(15); break;
    }
#undef HK_TM_LAUNCH
}
int test_majorant_stride() { return 1 + 3 * HK_TEST_MAJ_SEGS; }
void launch_test_trace_lean(hipStream_t s, int n_cu, const DScene& sc, int anyhit, int n, const float* o, const float* d, const float* tmax, float* t, int* prim, float* uv) {
    const int blocks = n_cu * 2;
#define HK_TL_LAUNCH(A, S) hipLaunchKernelGGL((k_test_trace_lean<A, S>), dim3(blocks), dim3(HK_TRACE_BLOCK), 0, s, sc, n, o, d, tmax, t, prim, uv)
    if (sc.bvh_depth <= 16) {
        if (anyhit) HK_TL_LAUNCH(true, 16); else HK_TL_LAUNCH(false, 16);
    } else if (sc.qnodes != nullptr) {   // the kernels' own choice for a deep tree: its quantised nodes
        if (anyhit) hipLaunchKernelGGL((k_test_trace_lean<true, HK_LDS_STACK, true>), dim3(blocks), dim3(HK_TRACE_BLOCK), 0, s, sc, n, o, d, tmax, t, prim, uv);
        else hipLaunchKernelGGL((k_test_trace_lean<false, HK_LDS_STACK, true>), dim3(blocks), dim3(HK_TRACE_BLOCK), 0, s, sc, n, o, d, tmax, t, prim, uv);
    } else {
        if (anyhit) HK_TL_LAUNCH(true, HK_LDS_STACK); else HK_TL_LAUNCH(false, HK_LDS_STACK);
    }
#undef HK_TL_LAUNCH
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
void launch_sobol_table(hipStream_t s, const DSobol& sob, const DFrame& fr, uint2* table, int rows) {
    hipLaunchKernelGGL(k_sobol_table, dim3(grid_for((long)rows * fr.n_pixels_padded > 0x3fffffff ? 0x3fffffff : rows * fr.n_pixels_padded, 256, 8192)), dim3(256), 0, s, sob, fr, table, rows);
}
void launch_sobol_lo_table(hipStream_t s, const DSobol& sob, const DFrame& fr, uint16_t* table, int rows, int base, int stride, int count) {
    hipLaunchKernelGGL(k_sobol_lo_table, dim3(65536), dim3(256), 0, s, sob, fr, table, rows, base, stride, count);
}
void launch_postprocess(hipStream_t s, const hk_postprocess_params& P, const float* src, const float* depth, float* dst, int h, int w) {
    hipLaunchKernelGGL(k_postprocess, dim3(grid_for(h * w, 256, 8192)), dim3(256), 0, s, P, src, depth, dst, h, w);
}
void launch_denoise_variance(hipStream_t s, const float* src, float* variance, int h, int w) {
    hipLaunchKernelGGL(k_denoise_variance, dim3(grid_for(h * w, 256, 8192)), dim3(256), 0, s, src, variance, h, w);
}
void launch_denoise_atrous(hipStream_t s, const hk_denoise_params& P, int step, const float* src, const float* normal, const float* depth, const float* variance,
                           float* dst, int h, int w) {
    hipLaunchKernelGGL(k_denoise_atrous, dim3(grid_for(h * w, 256, 8192)), dim3(256), 0, s, P, step, src, normal, depth, variance, dst, h, w);
}
void launch_aux(hipStream_t s, const DScene& sc, const DCamera& cam, int h, int w, float miss_depth, float* albedo, float* normal, float* depth) {
    hipLaunchKernelGGL(k_aux, dim3(grid_for(h * w, HK_TRACE_BLOCK, 2048)), dim3(HK_TRACE_BLOCK), 0, s, sc, cam, h, w, miss_depth, albedo, normal, depth);
}

}  // namespace hk

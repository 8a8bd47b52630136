// hk_api.cpp — host side of the C-ABI (include/hikari_mi355x.h): contexts, scene upload (BVH + light-BVH
// build, constant-colour baking), film, integrator state and the wavefront driver that replaces the
// reference's host loop `render!` (src/integrators/volpath/volpath.jl:445-636).  Unlike the reference
// loop (8-10 blocking `length(queue)` readbacks per bounce, workqueue.jl:108-111) nothing here reads
// the device back inside a frame: queue sizes stay in HBM and every kernel sizes itself from them.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <memory>
#include <mutex>

#include <dlfcn.h>
#include <string>
#include <unordered_map>
#include <vector>

#include "bvh_build.h"
#include "hikari_mi355x.h"
#include "hk_nanovdb.h"
#include "hk_types.h"

namespace hk {
void launch_camera(hipStream_t, int, const DPathState&, const DFrame&, const DTables&, const DFilter&, const DCamera&, const DSobol&, int);
void launch_trace(hipStream_t, int, const DPathState&, const DScene&, const DTables&, const DFrame&, int, DStats*);
void launch_shadow(hipStream_t, int, const DPathState&, const DScene&, const DTables&, const DFrame&, int, DStats*);
void launch_escaped(hipStream_t, int, const DPathState&, const DScene&, const DTables&, const DFrame&, int);
void launch_medium(hipStream_t, int, const DPathState&, const DScene&, const DTables&, const DFrame&, const DSobol&, int, DStats*);
void launch_detect_camera_medium(hipStream_t, const DPathState&, const DScene&, float, float, float, DStats*);
void launch_shade(hipStream_t, int, int, const DPathState&, const DScene&, const DTables&, const DFrame&, const DSobol&, int, int, DStats*);
bool grey_compact_ok(const DScene&);
bool preselect_lights(const DScene&, const DPathState&);
bool small_pass_fusable(const DScene&, uint32_t);
bool launch_small_pass(hipStream_t, int, const DPathState&, const DScene&, const DTables&, const DFrame&, const DFilter&, const DCamera&, const DSobol&, int, uint32_t, DStats*, bool, void*, int);
void launch_light_select(hipStream_t, int, const DPathState&, const DScene&, const DTables&, const DFrame&, const DSobol&, int, uint32_t, DStats*);
void launch_film(hipStream_t, const DPathState&, const DFrame&, const DTables&, void*, bool);
void launch_segment_lists(hipStream_t, const DPathState&, int, const int*, const int*);
void launch_finalize(hipStream_t, const void*, bool, float*, int, int);
void launch_test_trace(hipStream_t, const DScene&, int, const float*, const float*, const float*, float*, int*, float*);
void launch_test_sobol(hipStream_t, const DTables&, const DSobol&, int, const int*, const int*, const int*, const int*, float*, float*);
void launch_test_camera(hipStream_t, const DTables&, const DFilter&, const DCamera&, const DSobol&, int, int, const int*, const int*, const int*, float*);
void launch_test_uplift(hipStream_t, const DTables&, int, int, const float*, const float*, float*);
void launch_test_light_bvh(hipStream_t, const DScene&, int, const float*, const float*, const float*, int*, float*, const int*, float*);
void launch_aux(hipStream_t, const DScene&, const DCamera&, int, int, float, float*, float*, float*);
void launch_postprocess(hipStream_t, const hk_postprocess_params&, const float*, const float*, float*, int, int);
void launch_denoise_variance(hipStream_t, const float*, float*, int, int);
void launch_denoise_atrous(hipStream_t, const hk_denoise_params&, int, const float*, const float*, const float*, const float*, float*, int, int);
void launch_sobol_table(hipStream_t, const DSobol&, const DFrame&, uint2*, int);
void launch_sobol_lo_table(hipStream_t, const DSobol&, const DFrame&, uint16_t*, int, int, int, int);
void launch_test_light(hipStream_t, const DScene&, const DTables&, int, int, int, const float*, const float*, const float*, float*);
void launch_test_bsdf(hipStream_t, const DScene&, const DTables&, int, int, int, int, const float*, const float*, const float*, const float*, const float*, const float*, float*);
void launch_test_mix(hipStream_t, const DScene&, int, int, const float*, const float*, const float*, int*);
void launch_test_medium(hipStream_t, const DScene&, const DTables&, int, int, int, const float*, const float*, const float*, const float*, float*);
int test_majorant_stride();
void launch_test_trace_lean(hipStream_t, int, const DScene&, int, int, const float*, const float*, const float*, float*, int*, float*);
}  // namespace hk

static thread_local std::string g_err;
static int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}

// ---------------------------------------------------------------------------------------------------
// RUN-TIME KNOBS.  The library never calls getenv on a render path: hk_ctx_create copies the HK_* variables it knows from the
// environment ONCE into the context, hk_ctx_set_option changes one afterwards, and the code asks hk::knob("HK_X") — a lookup in the
// table of the context whose entry point is running on this thread (KnobScope).  A host that setenv()s beside a render is harmless.
// ---------------------------------------------------------------------------------------------------
namespace hk {
struct Knobs {
    std::unordered_map<std::string, std::string> kv;
};
static thread_local const Knobs* tl_knobs = nullptr;
const char* knob(const char* name) {
    if (!tl_knobs) return nullptr;
    auto it = tl_knobs->kv.find(name);
    return it == tl_knobs->kv.end() ? nullptr : it->second.c_str();
}
static const char* const KNOB_NAMES[] = {
    "HK_BATCH_PATHS_M", "HK_BVH_LEAF", "HK_BVH_BINS", "HK_QNODES", "HK_DEBUG_ALLOC", "HK_DELTA_ADVANCE", "HK_DYNAMIC_SEGMENTS", "HK_GREY", "HK_GREY_COMPACT", "HK_GREY_FLAT", "HK_MAX_PATHS_M",
    "HK_MID_LISTS", "HK_MID_PASS_PATHS_M", "HK_NODE_CACHE", "HK_NVDB_DENSE_MB", "HK_OVERLAP", "HK_PIPELINE", "HK_PIPELINE_AFTER", "HK_PIPELINE_MAX_PATHS_M", "HK_PRESELECT",
    "HK_SELECT_MIN_IDLE", "HK_SHADOW_FEED_ROUNDS", "HK_SHADOW_TRACK_BATCH", "HK_SMALL_PASS", "HK_SMALL_PASS_WAVES", "HK_SOBOL_LO_GB", "HK_SOBOL_TABLE_ONLY",
    "HK_STATE_CACHE_GB", "HK_STATE_SLAB", "HK_TICKET_SHARE", "HK_TRACK_ADVANCE", "HK_TRACK_EXTRA_ADVANCE", "HK_TRACK_MIN_PENDING", "HK_TRACK_POOL", "HK_TRACK_REFILL_IDLE",
    "HK_WALK_POOL", "HK_WALK_REFILL_IDLE", "HK_WALK_SPLIT", "HK_WAVES_PER_CU", "HK_READBACK_PIN", "HK_DEFER_EXTERNAL", "HK_SELECT_POOL", "HK_SMALL_PASS_FUSED", "HK_SMALL_PASS_MERGED", "HK_OCC_SCALE", "HK_ESCAPED_UNROLL", "HK_SHADOW_FINAL", "HK_TRI_PACK", "HK_LEAN_RECORDS"};
static bool known_knob(const char* name) {
    for (const char* k : KNOB_NAMES)
        if (std::strcmp(k, name) == 0) return true;
    return false;
}
}  // namespace hk
struct KnobScope {   // the knobs of `k` answer hk::knob on this thread until the scope ends (entry points nest: the outer one is restored)
    const hk::Knobs* prev;
    explicit KnobScope(const hk::Knobs* k) : prev(hk::tl_knobs) { hk::tl_knobs = k; }
    ~KnobScope() { hk::tl_knobs = prev; }
};
#define HIP_TRY(expr)                                                                                          \
    do {                                                                                                       \
        hipError_t e_ = (expr);                                                                                \
        if (e_ != hipSuccess) return fail(HK_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

namespace {
// PATH-STATE SLABS outlive the integrator that asked for them: the ~40 arrays of a path state are carved from one allocation, and a slab
// that is let go is kept (per device, up to HK_STATE_CACHE_GB = 128 in total, the smallest ones dropped first) for the next path state
// that fits it.  Two reasons, both measured on the cloud config (DESIGN.md §5 "two speeds"): allocating 84 GB takes 0.7 - 4 s, and where
// the driver places a LATER big allocation decides whether the gather-heavy kernels run 7 - 9 % slower for the life of that integrator.
// Everything cached is given back when a hipMalloc fails (then retried) and when a context of that device is destroyed.
struct SlabCache {
    struct Entry {
        void* p;
        size_t bytes;
        int dev;
    };
    std::mutex m;
    std::vector<Entry> free_list;
    size_t cap_bytes = (size_t)128 << 30;   // HK_STATE_CACHE_GB (process-wide: the last context created / option set decides)
    size_t cap() const { return cap_bytes; }
    size_t total(int dev) {
        std::lock_guard<std::mutex> g(m);
        size_t t = 0;
        for (const Entry& e : free_list) t += e.dev == dev ? e.bytes : 0;
        return t;
    }
    void* take(int dev, size_t need, size_t& got) {   // the smallest slab that holds `need` without being more than twice as large
        std::lock_guard<std::mutex> g(m);
        int best = -1;
        for (int i = 0; i < (int)free_list.size(); ++i) {
            const Entry& e = free_list[i];
            if (e.dev == dev && e.bytes >= need && e.bytes <= 2 * need + ((size_t)64 << 20) && (best < 0 || e.bytes < free_list[best].bytes)) best = i;
        }
        if (best < 0) return nullptr;
        void* p = free_list[best].p;
        got = free_list[best].bytes;
        free_list.erase(free_list.begin() + best);
        return p;
    }
    void give(int dev, void* p, size_t bytes) {
        std::vector<void*> drop;
        {
            std::lock_guard<std::mutex> g(m);
            free_list.push_back(Entry{p, bytes, dev});
            size_t t = 0;
            for (const Entry& e : free_list) t += e.bytes;
            const size_t limit = cap();
            while (!free_list.empty() && (t > limit || free_list.size() > 8)) {   // the smallest goes first: the big ones are the expensive ones
                int k = 0;
                for (int i = 1; i < (int)free_list.size(); ++i)
                    if (free_list[i].bytes < free_list[k].bytes) k = i;
                t -= free_list[k].bytes;
                drop.push_back(free_list[k].p);
                free_list.erase(free_list.begin() + k);
            }
        }
        for (void* q : drop) (void)hipFree(q);
    }
    void trim(int dev) {   // dev < 0: every device
        std::vector<void*> drop;
        {
            std::lock_guard<std::mutex> g(m);
            for (int i = (int)free_list.size() - 1; i >= 0; --i)
                if (dev < 0 || free_list[i].dev == dev) {
                    drop.push_back(free_list[i].p);
                    free_list.erase(free_list.begin() + i);
                }
        }
        for (void* q : drop) (void)hipFree(q);
    }
};
SlabCache g_slabs;
struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    int slab_dev = -1;   // >= 0: a path-state slab of that device (goes back to g_slabs, not to the driver)
    void release() {
        if (p && slab_dev >= 0) {
            // hipFree would have waited for the kernels that still use the memory; a slab goes back to the cache instead, so its OWNER
            // waits first — for the streams of its own context only (quiesce), not for every stream of the host application
            g_slabs.give(slab_dev, p, bytes);
        } else if (p)
            (void)hipFree(p);
        p = nullptr, bytes = 0, slab_dev = -1;
    }
    ~DevBuf() { release(); }
    hipError_t alloc(size_t n) {
        release();
        bytes = n;
        if (n == 0) return hipSuccess;
        hipError_t e = hipMalloc(&p, n);
        if (e != hipSuccess) {   // the cached slabs are memory too
            (void)hipGetLastError();
            g_slabs.trim(-1);
            e = hipMalloc(&p, n);
        }
        if (e != hipSuccess) p = nullptr, bytes = 0;
        return e;
    }
    hipError_t alloc_slab(int dev, size_t n) {   // a cached slab that fits, or a new one
        release();
        size_t got = 0;
        if (void* q = g_slabs.take(dev, n, got)) {
            p = q, bytes = got, slab_dev = dev;
            return hipSuccess;
        }
        const hipError_t e = alloc(n);
        if (e == hipSuccess) slab_dev = dev;
        return e;
    }
    hipError_t upload(const void* src, size_t n) {
        hipError_t e = alloc(n ? n : 4);
        if (e != hipSuccess) return e;
        if (n) e = hipMemcpy(p, src, n, hipMemcpyHostToDevice);
        return e;
    }
    template <class T>
    T* as() const {
        return reinterpret_cast<T*>(p);
    }
};
}  // namespace

struct hk_ctx {
    int device = 0;
    hk::Knobs knobs;                      // HK_* as of hk_ctx_create, then hk_ctx_set_option
    bool own_stream_order = true;         // false: the caller handed hk_ctx_create a stream of its own and may order work after a render with stream / event calls
    hipStream_t stream = nullptr;
    hipStream_t aux = nullptr;            // second stream: the shadow rays of bounce d run beside the traversal of bounce d + 1
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    int overlap = -1;                     // the shadow kernels on a second stream: -1 auto (scenes with a deep BVH), HK_OVERLAP=0 / 1 never / always
    int small_streak = 0;                 // consecutive small one-pass render calls so far (the lanes start after HK_PIPELINE_AFTER of them)
    int n_cu = 256;
    int waves_per_cu = 0;      // HK_WAVES_PER_CU: fixed number of wave segments per CU (0 = sized from the pass)
    int stat_rows = 8192;      // DStats rows, indexed by PHYSICAL wave: n_cu * 32 (8 waves x 4 SIMDs is the residency limit)
    DevBuf sobol, cie, r2s_scale, r2s_coeffs, r2s_points, stats;
    DTables tables{};
    bool have_tables = false;
    std::vector<float> h_r2s_scale, h_r2s_coeffs;
    hk::RGB2Spec r2s_host;
    int count_nodes = 0, time_kernels = 0;
    unsigned long long fused_passes = 0;   // passes rendered by k_small_pass (one launch)
    // timing
    std::vector<std::pair<hipEvent_t, hipEvent_t>> trace_events;   // class 0
    std::vector<std::pair<hipEvent_t, hipEvent_t>> class_events[6];  // 1 shadow, 2 shade, 3 other, 4 media, 5 light selection (reported inside the shade class AND on its own)
    uint64_t shadow_launches = 0, shade_launches = 0, media_launches = 0, select_launches = 0;
    std::vector<hipEvent_t> event_pool;
    hipEvent_t ev_begin = nullptr, ev_end = nullptr;
    bool have_span = false;
    double seconds_trace = 0.0, seconds_total = 0.0;
    uint64_t trace_launches = 0;
    DStats host_stats{};
    // PIPELINED SMALL PASSES.  A one-sample call (the reference's render!, volpath.jl:445-450: what an interactive viewer drives) puts
    // < 1 path per resident lane in flight: its ~45 launches are each bound by the latency of ONE wave's chunk (50 - 90 us at any
    // path count, profiles/r04_progressive_timeline.txt), the chip idles.  Such calls are independent of each other (another sample
    // index of the same scene), so consecutive small calls go to HK_PIPELINE (default 4) LANES in turn — a stream, a path-state set
    // and a statistics block each — and run beside each other; only the film kernels are chained (sums in call order: the film stays
    // bit-identical to the sequential one).  Whatever reads or rewrites film / state / scene on the context's stream joins the
    // lanes first (join_lanes).
    struct Lane {
        hipStream_t stream = nullptr;
        hipEvent_t done = nullptr;
        DevBuf stats;
    };
    enum { MAX_LANES = 16 };
    Lane lanes[MAX_LANES];        // (created on first use)
    int next_lane = 0;
    // SMALL RENDER CALLS ARE BATCHED (hk_render_tile): consecutive one-pass calls that continue each other — same scene, integrator, film,
    // camera, pixel range and stride, sample indices following on — are only noted here and rendered as ONE pass when something looks
    // (flush_pending: every entry point that reads or rewrites film / statistics / scene / integrator, hk_sync, a call that does not fit)
    struct Pending {
        bool active = false;
        hk_scene* sc = nullptr;
        hk_integrator* I = nullptr;
        hk_film* film = nullptr;
        hk_camera cam{};
        int first = 0, n = 0, stride = 1, x0 = 0, y0 = 0, x1 = 0, y1 = 0, calls = 0;
    } pending;
    bool lanes_dirty = false;     // a lane holds work the context's stream has not waited for
    bool film_chain = false;      // ev_film marks the last film kernel of a lane
    hipEvent_t ev_main = nullptr, ev_film = nullptr;
};

// every stream this context has launched on is idle afterwards (before path-state memory changes hands); other streams of the process are not touched
static void quiesce(hk_ctx* c) {
    if (!c) return;
    (void)hipStreamSynchronize(c->stream);
    if (c->aux) (void)hipStreamSynchronize(c->aux);
    for (auto& l : c->lanes)
        if (l.stream) (void)hipStreamSynchronize(l.stream);
}
// the context's stream waits for everything the lanes were given (cheap when nothing is pending)
static int flush_pending(hk_ctx* c);
static int join_lanes(hk_ctx* c) {
    if (!c) return HK_OK;
    if (int e = flush_pending(c)) return e;
    if (!c->lanes_dirty) return HK_OK;
    for (auto& l : c->lanes)
        if (l.done) HIP_TRY(hipStreamWaitEvent(c->stream, l.done, 0));
    c->lanes_dirty = false;
    c->film_chain = false;
    if (c->have_span) HIP_TRY(hipEventRecord(c->ev_end, c->stream));
    return HK_OK;
}

struct hk_scene {
    hk_ctx* ctx = nullptr;
    DevBuf nodes, qnodes, leaf_tris, positions, normals, uvs, tangents, meta, tri_shade, materials, textures, spectra, mis, lights, lnodes, trails, infinite;
    std::vector<DevBuf*> tex_data;
    std::vector<DevBuf*> spec_data;
    std::vector<DevBuf*> media_data;
    std::vector<DevBuf*> env_data;
    DevBuf envmaps;
    DevBuf media;
    DScene d{};
    uint32_t kinds_mask = 0;
    int n_materials = 0;
    int bvh_nodes = 0, bvh_leaf_tris = 0, bvh_depth = 0;
    hk::LightBVH lbvh;
    ~hk_scene() {
        for (auto* b : tex_data) delete b;
        for (auto* b : spec_data) delete b;
        for (auto* b : media_data) delete b;
        for (auto* b : env_data) delete b;
    }
};

struct hk_film {
    hk_ctx* ctx = nullptr;
    int width = 0, height = 0;
    bool f64 = false;
    DevBuf own;
    void* accum = nullptr;  // device
    bool external = false;  // the caller owns `accum` (and may read it behind stream / event ordering of its own)
    bool exposed = false;   // hk_film_accum_device_ptr handed the accumulators out: the caller may keep the pointer, so calls into this film are never only noted
    DevBuf readback;
    // hk_film_read_rgb / _async: the finalized frame lands in PINNED host memory (two buffers in turn), or straight in the caller's
    // buffer when the caller named it with hk_film_pin_host (HK_READBACK_PIN=1: also a pointer that has come twice in a row)
    float* staging[2] = {nullptr, nullptr};
    int staging_next = 0, staging_last = -1;   // which buffer the next async read fills / the last one filled
    bool read_in_flight = false;
    hipEvent_t ev_read = nullptr;
    void* last_out = nullptr;       // the caller's buffer of the previous synchronous read
    void* pinned_user = nullptr;    // ... registered with the driver (hipHostRegister) — the copy goes there directly
    bool pinned_explicit = false;   // registered by hk_film_pin_host (stays until hk_film_unpin_host / hk_film_destroy)
    void* pin_failed = nullptr;     // HK_READBACK_PIN=1: the pointer whose registration the driver refused (not retried every frame)
};

struct hk_integrator {
    hk_ctx* ctx = nullptr;
    hk_integrator_params p{};
    DFilter filter{};
    DevBuf f_func, f_mcdf, f_mfunc, f_ccdf;
    // path state (the reference's VolPathState, volpath-state.jl:29-181)
    DPathState st{};
    std::vector<DevBuf*> bufs;
    int st_capacity = 0, st_depth = 0, st_media = -1;   // what the retained path state was allocated for
    bool mid_pass = false;                              // the current pass is a mid-size one of a closed scene (ensure_state): static stride, one stream
    int slab_mode = 0;         // 0: one allocation per array; 1: measuring the slab; 2: carving it
    void* slab_base = nullptr;
    size_t slab_off = 0;
    DevBuf sobol_table;  // DSobol::hi_table
    int sobol_rows = 0, sobol_stride = 0, sobol_log2 = -1, sobol_digits = -1, sobol_x0 = -1, sobol_y0 = -1, sobol_tiles_x = -1;
    DevBuf sobol_lo;     // DSobol::lo_table
    int lo_rows = 0, lo_base = -1, lo_sample_stride = -1, lo_count = 0;
    // the path-state sets of the context's lanes (pipelined small passes); a render on lane l swaps set l in for the duration of the call
    struct StateSet {
        DPathState st{};
        std::vector<DevBuf*> bufs;
        int st_capacity = 0, st_depth = 0, st_media = -1;
    };
    std::vector<StateSet> lane_sets;
    void swap_set(StateSet& o) {
        std::swap(st, o.st);
        std::swap(bufs, o.bufs);
        std::swap(st_capacity, o.st_capacity);
        std::swap(st_depth, o.st_depth);
        std::swap(st_media, o.st_media);
    }
    ~hk_integrator() {
        for (auto* b : bufs) delete b;
        for (auto& ls : lane_sets)
            for (auto* b : ls.bufs) delete b;
    }
};

// ---------------------------------------------------------------------------------------------------
extern "C" const char* hk_last_error(void) { return g_err.c_str(); }

extern "C" int32_t hk_ctx_create(int32_t device_id, void* stream, hk_ctx** out) {
    if (!out) return fail(HK_ERR_INVALID, "out is null");
    int n = 0;
    HIP_TRY(hipGetDeviceCount(&n));
    if (device_id < 0 || device_id >= n) return fail(HK_ERR_INVALID, "bad device id");
    HIP_TRY(hipSetDevice(device_id));
    hk_ctx* c = new hk_ctx();
    c->device = device_id;
    c->stream = (hipStream_t)stream;
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device_id));
    c->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    c->own_stream_order = stream == nullptr;
    for (const char* name : hk::KNOB_NAMES)   // the ONLY place the library reads the environment (std::getenv below: nowhere else)
        if (const char* e = std::getenv(name)) c->knobs.kv[name] = e;
    KnobScope knobs(&c->knobs);
    {
        if (const char* e = hk::knob("HK_WAVES_PER_CU")) c->waves_per_cu = std::atoi(e) > 0 ? std::atoi(e) : 0;
        c->stat_rows = c->n_cu * 32 * 2;   // second half: the kernels of the second stream (their waves have the same physical ids)
        if (const char* e = hk::knob("HK_OVERLAP")) c->overlap = std::atoi(e) ? 1 : 0;
        else c->overlap = -1;
        if (const char* e = hk::knob("HK_STATE_CACHE_GB"))
            if (std::atol(e) >= 0) g_slabs.cap_bytes = (size_t)std::atol(e) << 30;
    }
    {
        std::vector<DStats> zero((size_t)c->stat_rows);
        std::memset(zero.data(), 0, zero.size() * sizeof(DStats));
        HIP_TRY(c->stats.upload(zero.data(), zero.size() * sizeof(DStats)));
    }
    HIP_TRY(hipEventCreate(&c->ev_begin));
    HIP_TRY(hipEventCreate(&c->ev_end));
    *out = c;
    return HK_OK;
}
extern "C" int32_t hk_ctx_destroy(hk_ctx* c) {
    if (!c) return HK_OK;
    (void)hipSetDevice(c->device);
    int status = join_lanes(c);
    if (hipStreamSynchronize(c->stream) != hipSuccess && status == HK_OK) status = fail(HK_ERR_DEVICE, "hk_ctx_destroy: the context's stream reports an error");
    for (auto& l : c->lanes) {
        if (l.stream) (void)hipStreamSynchronize(l.stream);
        if (l.done) (void)hipEventDestroy(l.done);
        if (l.stream) (void)hipStreamDestroy(l.stream);
    }
    if (c->ev_main) (void)hipEventDestroy(c->ev_main);
    if (c->ev_film) (void)hipEventDestroy(c->ev_film);
    for (auto& e : c->trace_events) {
        (void)hipEventDestroy(e.first);
        (void)hipEventDestroy(e.second);
    }
    for (auto e : c->event_pool) (void)hipEventDestroy(e);
    if (c->ev_begin) (void)hipEventDestroy(c->ev_begin);
    if (c->ev_end) (void)hipEventDestroy(c->ev_end);
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->ev_join) (void)hipEventDestroy(c->ev_join);
    if (c->aux) (void)hipStreamDestroy(c->aux);
    g_slabs.trim(c->device);
    delete c;
    return status;
}
extern "C" int32_t hk_ctx_set_option(hk_ctx* c, const char* name, const char* value) {
    if (!c || !name) return fail(HK_ERR_INVALID, "null argument");
    if (!hk::known_knob(name)) return fail(HK_ERR_INVALID, std::string("unknown option ") + name);
    HIP_TRY(hipSetDevice(c->device));
    if (int e = join_lanes(c)) return e;   // calls that were only noted are rendered under the options they were made with
    if (value) c->knobs.kv[name] = value;
    else c->knobs.kv.erase(name);
    if (std::strcmp(name, "HK_STATE_CACHE_GB") == 0) {
        g_slabs.cap_bytes = (size_t)(value && std::atol(value) >= 0 ? std::atol(value) : 128) << 30;
        if (g_slabs.cap_bytes == 0) g_slabs.trim(c->device);
    }
    return HK_OK;
}
extern "C" int32_t hk_ctx_get_option(hk_ctx* c, const char* name, char* out, int32_t out_bytes) {
    if (!c || !name || (out_bytes > 0 && !out)) return fail(HK_ERR_INVALID, "null argument");
    if (!hk::known_knob(name)) return fail(HK_ERR_INVALID, std::string("unknown option ") + name);
    auto it = c->knobs.kv.find(name);
    if (it == c->knobs.kv.end()) {   // unset: the built-in default applies — HK_UNSET, not an error code (HK_ERR_INVALID is -1: a typo must not read as "default")
        if (out_bytes > 0) out[0] = 0;
        return HK_UNSET;
    }
    if (out_bytes > 0) std::snprintf(out, (size_t)out_bytes, "%s", it->second.c_str());
    return (int32_t)it->second.size();
}
extern "C" int32_t hk_flush(hk_ctx* c) {
    if (!c) return fail(HK_ERR_INVALID, "null ctx");
    HIP_TRY(hipSetDevice(c->device));
    return join_lanes(c);   // noted calls are enqueued on the context's stream (and the lanes joined): stream-ordered from here on
}
extern "C" int32_t hk_trim_cache(hk_ctx* c) {
    if (!c) return fail(HK_ERR_INVALID, "null ctx");
    HIP_TRY(hipSetDevice(c->device));
    g_slabs.trim(c->device);
    return HK_OK;
}
extern "C" int32_t hk_ctx_set_tables(hk_ctx* c, const hk_tables* t) {
    if (!c || !t || !t->sobol_matrices || t->sobol_count < 104 || !t->cie_x || !t->rgb2spec_coeffs) return fail(HK_ERR_INVALID, "bad tables");
    HIP_TRY(hipSetDevice(c->device));
    if (int e = flush_pending(c)) return e;   // (noted small calls are rendered with the tables they were made under)
    if (c->lanes_dirty) {   // renders in flight on the lanes read the tables that are about to be replaced
        if (int e = join_lanes(c)) return e;
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
    HIP_TRY(c->sobol.upload(t->sobol_matrices, 104 * sizeof(uint32_t)));  // only Sobol dims 0,1 are ever read
    std::vector<float> cie(3 * 471);
    std::memcpy(&cie[0], t->cie_x, 471 * 4);
    std::memcpy(&cie[471], t->cie_y, 471 * 4);
    std::memcpy(&cie[942], t->cie_z, 471 * 4);
    HIP_TRY(c->cie.upload(cie.data(), cie.size() * 4));
    int res = t->rgb2spec_res;
    size_t nco = (size_t)3 * res * res * res * 3;
    c->h_r2s_scale.assign(t->rgb2spec_scale, t->rgb2spec_scale + res);
    c->h_r2s_coeffs.assign(t->rgb2spec_coeffs, t->rgb2spec_coeffs + nco);
    c->r2s_host.res = res;
    c->r2s_host.scale = c->h_r2s_scale.data();
    c->r2s_host.coeffs = c->h_r2s_coeffs.data();
    HIP_TRY(c->r2s_scale.upload(t->rgb2spec_scale, res * 4));
    HIP_TRY(c->r2s_coeffs.upload(t->rgb2spec_coeffs, nco * 4));
    {   // corner-per-load copy of the coefficient table and the monotonicity the device's cell search relies on
        const size_t R = (size_t)res;
        std::vector<float> pts(3 * R * R * R * 4);
        for (size_t m = 0; m < 3; ++m)
            for (size_t z = 0; z < R; ++z)
                for (size_t y = 0; y < R; ++y)
                    for (size_t x = 0; x < R; ++x) {
                        float* o = &pts[((((m * R + z) * R + y) * R) + x) * 4];
                        for (size_t k = 0; k < 3; ++k) o[k] = t->rgb2spec_coeffs[m + 3 * (z + R * (y + R * (x + R * k)))];
                        o[3] = 0.0f;
                    }
        HIP_TRY(c->r2s_points.upload(pts.data(), pts.size() * 4));
        bool sorted = true;
        for (int i = 1; i < res; ++i)
            if (!(t->rgb2spec_scale[i - 1] <= t->rgb2spec_scale[i])) sorted = false;
        c->tables.rgb2spec_sorted = sorted ? 1 : 0;
    }
    // Sobol dims 0/1 have closed forms (hk_device.h sobol_matrix_product); use them only if the caller's table
    // really is that matrix, otherwise keep the table loop.
    bool closed = true;
    {
        uint32_t col = 0x80000000u;
        for (int b = 0; b < 52; ++b) {
            uint32_t d0 = b < 32 ? (0x80000000u >> b) : 0u;
            if (t->sobol_matrices[b] != d0) closed = false;
            if (b % 32 == 0) col = 0x80000000u;
            if (t->sobol_matrices[52 + b] != col) closed = false;
            col ^= col >> 1;
        }
    }
    c->tables.sobol = closed ? nullptr : c->sobol.as<uint32_t>();
    c->tables.cie = c->cie.as<float>();
    c->tables.rgb2spec_scale = c->r2s_scale.as<float>();
    c->tables.rgb2spec_coeffs = c->r2s_coeffs.as<float>();
    c->tables.rgb2spec_res = res;
    c->tables.rgb2spec_points = c->r2s_points.as<float4>();
    c->have_tables = true;
    return HK_OK;
}

// ---- constant-colour baking (host float32 arithmetic identical to the device/run-time path) ----------
namespace {
inline float clampf(float v, float lo, float hi) { return v < lo ? lo : (v > hi ? hi : v); }
inline float max3(float a, float b, float c) {
    float m = a > b ? a : b;
    return m > c ? m : c;
}
void bake_bounded(const hk::RGB2Spec& t, float r, float g, float b, float out[4]) {
    hk::rgb_to_coeffs(t, r, g, b, out);
    out[3] = 1.0f;
}
void bake_unbounded(const hk::RGB2Spec& t, float r, float g, float b, float out[4]) {  // uplift_rgb_unbounded (uplift.jl:286-308)
    float m = max3(r, g, b);
    if (m <= 0.0f) {
        out[0] = out[1] = out[2] = out[3] = 0.0f;
        return;
    }
    hk::rgb_to_coeffs(t, r / m, g / m, b / m, out);
    out[3] = m / hk::poly_max(out);
}
void bake_illuminant(const hk::RGB2Spec& t, float r, float g, float b, float out[4]) {  // uplift.jl:514-538
    float m = max3(r, g, b);
    if (m <= 0.0f) {
        out[0] = out[1] = out[2] = out[3] = 0.0f;
        return;
    }
    float scale = 2.0f * m;
    hk::rgb_to_coeffs(t, r / scale, g / scale, b / scale, out);
    out[3] = scale;
}
enum { BAKE_BOUNDED, BAKE_BOUNDED_CLAMP, BAKE_UNBOUNDED };
int bake_mode(int kind, int slot) {
    if (kind == HK_MAT_MATTE && slot == 0) return BAKE_BOUNDED_CLAMP;  // clamp(kd_rgb) then uplift_rgb
    if ((kind == HK_MAT_CONDUCTOR || kind == HK_MAT_COATED_CONDUCTOR) && slot < 2) return BAKE_UNBOUNDED;
    return BAKE_BOUNDED;
}
}  // namespace


namespace {
// Range checks of a scene description: texture / spectrum / Mix child / area-light / medium indices.
std::string validate_desc(const hk_scene_desc& d) {
    auto need = [](int n, const void* p) { return n <= 0 || p != nullptr; };
    if (d.n_materials < 0 || d.n_textures < 0 || d.n_media_interfaces < 0 || d.n_lights < 0 || d.n_envmaps < 0 || d.n_media < 0 || d.n_spectra < 0) return "negative count";
    if (!need(d.n_materials, d.materials) || !need(d.n_textures, d.textures) || !need(d.n_media_interfaces, d.media_interfaces) || !need(d.n_lights, d.lights) ||
        !need(d.n_media, d.media) || !need(d.n_spectra, d.spectra))
        return "count > 0 with a null array";
    auto tex_ok = [&](int t, int channels) { return t < 0 || (t < d.n_textures && (channels == 0 || d.textures[t].channels == channels || d.textures[t].kind == 1)); };
    for (int i = 0; i < d.n_textures; ++i) {
        const hk_texture& t = d.textures[i];
        if (!t.data || t.width <= 0 || t.height <= 0 || (t.channels != 1 && t.channels != 4) || (t.kind != 0 && t.kind != 1)) return "bad texture record " + std::to_string(i);
        if (t.kind == 1 && (t.height != 3 || t.channels != 4)) return "vertex-colour texture must be face_colors[3, n_faces] RGBA";
    }
    for (int i = 0; i < d.n_spectra; ++i)
        if (d.spectra[i].n < 1 || !d.spectra[i].lambdas || !d.spectra[i].values) return "bad spectrum record " + std::to_string(i);
    for (int i = 0; i < d.n_materials; ++i) {
        const hk_material& m = d.materials[i];
        for (int k = 0; k < 4; ++k)
            if (!tex_ok(m.rgb[k].tex, 0)) return "material " + std::to_string(i) + ": rgb texture index out of range";
        for (int k = 0; k < 8; ++k)
            if (!tex_ok(m.f[k].tex, 0)) return "material " + std::to_string(i) + ": float texture index out of range";
        const bool reads_spectra = m.kind == HK_MAT_CONDUCTOR || m.kind == HK_MAT_COATED_CONDUCTOR;
        for (int k = 0; k < 2; ++k)
            if (reads_spectra && m.spectrum[k] >= d.n_spectra) return "material " + std::to_string(i) + ": spectrum index out of range";
        if (m.kind == HK_MAT_MIX && (m.i[0] < 0 || m.i[0] >= d.n_materials || m.i[1] < 0 || m.i[1] >= d.n_materials))
            return "MixMaterial " + std::to_string(i) + ": child material index out of range";
    }
    for (int i = 0; i < d.n_media_interfaces; ++i) {
        const hk_medium_interface& mi = d.media_interfaces[i];
        if (mi.material < 0 || mi.material >= d.n_materials) return "medium interface references a missing material";
        if (mi.inside < -1 || mi.inside >= d.n_media || mi.outside < -1 || mi.outside >= d.n_media) return "medium interface references a missing medium";
    }
    for (int i = 0; i < d.n_lights; ++i) {
        const hk_light& l = d.lights[i];
        if (l.kind < HK_LIGHT_POINT || l.kind > HK_LIGHT_DIFFUSE_AREA) return "unknown light kind";
        if (l.kind == HK_LIGHT_DIFFUSE_AREA && !tex_ok(l.Le.tex, 0)) return "area light: Le texture index out of range";
    }
    for (int t = 0; t < d.n_triangles; ++t) {
        const hk_tri_meta& m = d.meta[t];
        if ((int64_t)m.medium_interface_idx >= d.n_media_interfaces) return "triangle references a missing medium interface";
        if ((int64_t)m.arealight_flat_idx_1based > d.n_lights) return "triangle references a missing area light";
        // Q34: the index is the light COUNT at push time (scene-mesh.jl:127-128) while the flat order groups the lights by type
        // (light-sampler.jl:289-329): a light of an already-seen type pushed later shifts the area lights and the triangle then names
        // another DiffuseAreaLight — or a light of another kind, for which arealight_Le is black (diffuse-area.jl:81).  Accepted as is.
    }
    return std::string();
}
}  // namespace

extern "C" int32_t hk_scene_create(hk_ctx* c, const hk_scene_desc* d, hk_scene** out) {
    if (!c || !d || !out) return fail(HK_ERR_INVALID, "null argument");
    if (!c->have_tables) return fail(HK_ERR_INVALID, "hk_ctx_set_tables must be called first");
    KnobScope knobs(&c->knobs);
    if (d->n_envmaps > 0 && !d->envmaps) return fail(HK_ERR_INVALID, "n_envmaps > 0 but envmaps is null");
    for (int i = 0; i < d->n_lights; ++i)
        if (d->lights[i].kind == HK_LIGHT_ENVIRONMENT && (d->lights[i].envmap < 0 || d->lights[i].envmap >= d->n_envmaps))
            return fail(HK_ERR_INVALID, "environment light refers to a missing envmap");
    if (d->n_triangles < 0 || (d->n_triangles > 0 && (!d->positions || !d->meta))) return fail(HK_ERR_INVALID, "bad triangle arrays");
    {   // every index the device will follow is checked here: a malformed description is an error, never an out-of-bounds read
        std::string bad = validate_desc(*d);
        if (!bad.empty()) return fail(HK_ERR_INVALID, bad);
    }
    HIP_TRY(hipSetDevice(c->device));
    // the scene is owned by `guard` until it is handed to the caller: every error return below frees what was built so far
    std::unique_ptr<hk_scene> guard(new hk_scene());
    hk_scene* s = guard.get();
    s->ctx = c;
    const int T = d->n_triangles;
    // ---- classify surfaces: opaque = no medium transition and no alpha test possible ----
    std::vector<uint8_t> mat_alpha(d->n_materials > 0 ? d->n_materials : 1, 0);
    for (int i = 0; i < d->n_materials; ++i) {
        const hk_material& m = d->materials[i];
        if (m.kind == HK_MAT_MATTE && (m.rgb[0].tex >= 0 || m.rgb[0].c[3] < 1.0f)) mat_alpha[i] = 1;
    }
    bool all_opaque = true;
    std::vector<uint8_t> mi_opaque(d->n_media_interfaces > 0 ? d->n_media_interfaces : 1, 1);
    for (int i = 0; i < d->n_media_interfaces; ++i) {
        const hk_medium_interface& mi = d->media_interfaces[i];
        if (mi.material < 0 || mi.material >= d->n_materials) {
            return fail(HK_ERR_INVALID, "medium interface references a missing material");
        }
        bool op = mi.inside == mi.outside && !mat_alpha[mi.material];
        mi_opaque[i] = op;
        if (!op) all_opaque = false;
    }
    // ---- BVH ----
    hk::BVH bvh;
    {
        const char* leaf = hk::knob("HK_BVH_LEAF");
        const char* bins = hk::knob("HK_BVH_BINS");
        hk::build_bvh(d->positions, T, bvh, leaf ? std::atoi(leaf) : 4, bins ? std::atoi(bins) : 32);
    }
    s->bvh_nodes = (int)bvh.nodes.size();
    s->bvh_leaf_tris = (int)bvh.leaf_prims.size();
    s->bvh_depth = bvh.max_depth;
    std::vector<DNode> dn(bvh.nodes.size() ? bvh.nodes.size() : 1);
    for (size_t i = 0; i < bvh.nodes.size(); ++i) {
        const hk::BVHNode& n = bvh.nodes[i];
        dn[i].a = make_float4(n.lo0[0], n.hi0[0], n.lo0[1], n.hi0[1]);
        dn[i].b = make_float4(n.lo0[2], n.hi0[2], n.lo1[0], n.hi1[0]);
        dn[i].c = make_float4(n.lo1[1], n.hi1[1], n.lo1[2], n.hi1[2]);
        dn[i].c0 = n.c0;
        dn[i].c1 = n.c1;
        dn[i].pad0 = dn[i].pad1 = 0;
    }
    std::vector<float4> lt(3 * (bvh.leaf_prims.size() ? bvh.leaf_prims.size() : 1));
    for (size_t i = 0; i < bvh.leaf_prims.size(); ++i) {
        int prim = bvh.leaf_prims[i];
        const float* p = d->positions + 9 * (size_t)prim;
        uint32_t flags = 0;
        uint32_t mi = d->meta[prim].medium_interface_idx;
        if ((int)mi >= d->n_media_interfaces) {
            return fail(HK_ERR_INVALID, "triangle references a missing medium interface");
        }
        if (mi_opaque[mi]) flags |= HK_TRI_OPAQUE;
        float pw, fw;
        std::memcpy(&pw, &prim, 4);
        std::memcpy(&fw, &flags, 4);
        lt[3 * i + 0] = make_float4(p[0], p[1], p[2], pw);
        lt[3 * i + 1] = make_float4(p[3] - p[0], p[4] - p[1], p[5] - p[2], fw);  // e1 = v1 - v0
        lt[3 * i + 2] = make_float4(p[6] - p[0], p[7] - p[1], p[8] - p[2], 0.0f);  // e2 = v2 - v0
    }
    HIP_TRY(s->nodes.upload(dn.data(), dn.size() * sizeof(DNode)));
    // ---- quantised copy of a deep tree (DQNode, hk_types.h): 16-bit planes on a grid over the root box, lo rounded down and hi rounded
    //      up with one more cell of margin; read by the lean traversal kernels' deep-tree instantiations (HK_QNODES=0: off) ----
    float q_base[3] = {0, 0, 0}, q_cell[3] = {1, 1, 1};
    bool have_qnodes = false;
    {
        const char* qk = hk::knob("HK_QNODES");
        // (shallow trees gain nothing: their any-hit kernel with the nodes beyond its LDS cache read from a quantised array — measured,
        //  Cornell shadow class + 1.7 %, two-spheres + 0.7 % — is bound by instruction issue, and the conversions are instructions)
        if (bvh.max_depth > 16 && !bvh.nodes.empty() && (!qk || std::atoi(qk) != 0)) {
            for (int k = 0; k < 3; ++k) {
                const double ext = (double)bvh.hi[k] - (double)bvh.lo[k];
                q_cell[k] = (float)std::max(ext / 65527.0, 1e-30);
                q_base[k] = (float)((double)bvh.lo[k] - 3.0 * (double)q_cell[k]);
            }
            std::vector<DQNode> qn(bvh.nodes.size());
            auto quant = [&](float v, int axis, bool upper) -> uint32_t {
                if (!std::isfinite(v)) return upper ? 0u : 65535u;   // an empty box stays empty (lo > hi)
                const double g = ((double)v - (double)q_base[axis]) / (double)q_cell[axis];
                const double q = upper ? std::ceil(g) + 1.0 : std::floor(g) - 1.0;
                return (uint32_t)std::min(std::max(q, 0.0), 65535.0);
            };
            bool ok = true;
            for (size_t i = 0; i < bvh.nodes.size(); ++i) {
                const hk::BVHNode& n = bvh.nodes[i];
                for (int c = 0; c < 2; ++c)
                    for (int k = 0; k < 3; ++k) {
                        const float lo = c ? n.lo1[k] : n.lo0[k], hi = c ? n.hi1[k] : n.hi0[k];
                        const uint32_t ql = quant(lo, k, false), qh = quant(hi, k, true);
                        // the grid must contain the box with its margin (a box outside the root's bounds would be clipped: never for a child of the root)
                        if (std::isfinite(lo) && std::isfinite(hi) && ((double)q_base[k] + ql * (double)q_cell[k] > lo || (double)q_base[k] + qh * (double)q_cell[k] < hi)) ok = false;
                        qn[i].w[3 * c + k] = ql | (qh << 16);
                    }
                qn[i].c0 = n.c0;
                qn[i].c1 = n.c1;
            }
            if (ok) {
                HIP_TRY(s->qnodes.upload(qn.data(), qn.size() * sizeof(DQNode)));
                have_qnodes = true;
            }
        }
    }
    HIP_TRY(s->leaf_tris.upload(lt.data(), lt.size() * sizeof(float4)));
    HIP_TRY(s->positions.upload(d->positions, (size_t)T * 9 * 4));
    if (d->normals) HIP_TRY(s->normals.upload(d->normals, (size_t)T * 9 * 4));
    if (d->uvs) HIP_TRY(s->uvs.upload(d->uvs, (size_t)T * 6 * 4));
    if (d->tangents) HIP_TRY(s->tangents.upload(d->tangents, (size_t)T * 9 * 4));
    static_assert(sizeof(DTriMeta) == sizeof(hk_tri_meta), "meta layout");
    HIP_TRY(s->meta.upload(d->meta, (size_t)T * sizeof(hk_tri_meta)));
    // scenes whose attribute arrays (108 B per triangle) fit in an XCD's L2 gain nothing and pay the second address path (cloud config,
    // 24 triangles: shade + 4 %): records from 32 768 triangles up.  HK_TRI_PACK=0: never, 1: always (A/B switch and tests; films bit-identical)
    bool tri_pack = T >= 32768;
    if (const char* e = hk::knob("HK_TRI_PACK")) tri_pack = T > 0 && std::atoi(e) != 0;
    if (tri_pack) {   // packed shading records (DScene::tri_shade): p[9] n[9] uv[6] meta[3] pad[5] per triangle, one 128-byte line
        std::vector<float> rec((size_t)T * 32, 0.0f);
        const float nan = std::numeric_limits<float>::quiet_NaN();
        static const float default_uv[6] = {0.0f, 0.0f, 1.0f, 0.0f, 1.0f, 1.0f};   // (0,0), (1,0), (1,1): uv_at's defaults
        for (int t = 0; t < T; ++t) {
            float* r = rec.data() + (size_t)t * 32;
            std::memcpy(r, d->positions + 9 * (size_t)t, 36);
            if (d->normals) std::memcpy(r + 9, d->normals + 9 * (size_t)t, 36);
            else for (int k = 0; k < 9; ++k) r[9 + k] = nan;
            std::memcpy(r + 18, d->uvs ? d->uvs + 6 * (size_t)t : default_uv, 24);
            std::memcpy(r + 24, &d->meta[t], 12);
        }
        HIP_TRY(s->tri_shade.upload(rec.data(), rec.size() * sizeof(float)));
    }
    // ---- textures / spectra ----
    std::vector<DTexture> dt(d->n_textures > 0 ? d->n_textures : 1);
    for (int i = 0; i < d->n_textures; ++i) {
        const hk_texture& t = d->textures[i];
        DevBuf* b = new DevBuf();
        s->tex_data.push_back(b);
        HIP_TRY(b->upload(t.data, (size_t)t.width * t.height * t.channels * 4));
        dt[i].data = b->as<float>();
        dt[i].width = t.width;
        dt[i].height = t.height;
        dt[i].channels = t.channels;
        dt[i].pad = t.kind;  // 0 image, 1 vertex colours
    }
    HIP_TRY(s->textures.upload(dt.data(), dt.size() * sizeof(DTexture)));
    std::vector<DPLSpectrum> dsp(d->n_spectra > 0 ? d->n_spectra : 1);
    for (int i = 0; i < d->n_spectra; ++i) {
        const hk_pl_spectrum& sp = d->spectra[i];
        DevBuf* b = new DevBuf();
        s->spec_data.push_back(b);
        std::vector<float> both(2 * (size_t)sp.n);
        std::memcpy(both.data(), sp.lambdas, sp.n * 4);
        std::memcpy(both.data() + sp.n, sp.values, sp.n * 4);
        HIP_TRY(b->upload(both.data(), both.size() * 4));
        dsp[i].lambdas = b->as<float>();
        dsp[i].values = b->as<float>() + sp.n;
        dsp[i].n = sp.n;
        dsp[i].pad = 0;
    }
    HIP_TRY(s->spectra.upload(dsp.data(), dsp.size() * sizeof(DPLSpectrum)));
    // ---- materials: bake constant colours into sigmoid coefficients ----
    std::vector<DMaterial> dm(d->n_materials > 0 ? d->n_materials : 1);
    std::memset(dm.data(), 0, dm.size() * sizeof(DMaterial));
    for (int i = 0; i < d->n_materials; ++i) {
        const hk_material& m = d->materials[i];
        DMaterial& o = dm[i];
        o.kind = (m.kind >= 0 && m.kind <= HK_MAT_FALLBACK) ? m.kind : HK_MAT_FALLBACK;
        o.flags = m.flags;
        std::memcpy(o.i, m.i, sizeof o.i);
        std::memcpy(o.spectrum, m.spectrum, sizeof o.spectrum);
        std::memcpy(o.mix_key, m.mix_key, sizeof o.mix_key);
        for (int k = 0; k < 4; ++k) {
            DSpectrumParam& sp = o.rgb[k];
            sp.tex = m.rgb[k].tex;
            std::memcpy(sp.rgba, m.rgb[k].c, 16);
            float cf[4] = {0, 0, 0, 0};
            if (sp.tex < 0) {
                float r = m.rgb[k].c[0], g = m.rgb[k].c[1], b = m.rgb[k].c[2];
                if (o.kind == HK_MAT_DIFFUSE_TRANSMISSION && k < 2) {  // clamp(rgb * scale, 0, 1) (spectral-eval.jl:2098-2103)
                    const float sc = m.f[0].v;
                    r = clampf(r * sc, 0.0f, 1.0f), g = clampf(g * sc, 0.0f, 1.0f), b = clampf(b * sc, 0.0f, 1.0f);
                }
                if (o.kind == HK_MAT_COATED_CONDUCTOR && k == 2)  // reflectance mode: clamp(r, 0, 0.9999) (:2924-2928)
                    r = clampf(r, 0.0f, 0.9999f), g = clampf(g, 0.0f, 0.9999f), b = clampf(b, 0.0f, 0.9999f);
                switch (bake_mode(o.kind, k)) {
                    case BAKE_BOUNDED_CLAMP:
                        r = clampf(r, 0.0f, INFINITY);
                        g = clampf(g, 0.0f, INFINITY);
                        b = clampf(b, 0.0f, INFINITY);
                        bake_bounded(c->r2s_host, r, g, b, cf);
                        break;
                    case BAKE_UNBOUNDED: bake_unbounded(c->r2s_host, r, g, b, cf); break;
                    default: bake_bounded(c->r2s_host, r, g, b, cf); break;
                }
            }
            sp.coef = make_float4(cf[0], cf[1], cf[2], cf[3]);
        }
        for (int k = 0; k < 8; ++k) {
            o.f[k] = m.f[k].v;
            o.ftex[k] = m.f[k].tex;
        }
        if (o.kind != HK_MAT_MIX) s->kinds_mask |= 1u << o.kind;
    }
    s->n_materials = d->n_materials;
    HIP_TRY(s->materials.upload(dm.data(), dm.size() * sizeof(DMaterial)));
    static_assert(sizeof(DMediumInterface) == 16, "mi layout");
    std::vector<DMediumInterface> dmi(d->n_media_interfaces > 0 ? d->n_media_interfaces : 1);
    for (int i = 0; i < d->n_media_interfaces; ++i) dmi[i] = DMediumInterface{d->media_interfaces[i].material, d->media_interfaces[i].inside, d->media_interfaces[i].outside, 0};
    HIP_TRY(s->mis.upload(dmi.data(), dmi.size() * sizeof(DMediumInterface)));
    // ---- lights ----
    std::vector<DLight> dl(d->n_lights > 0 ? d->n_lights : 1);
    std::memset(dl.data(), 0, dl.size() * sizeof(DLight));
    int has_escape = 0, textured_emitters = 0;
    for (int i = 0; i < d->n_lights; ++i) {
        const hk_light& l = d->lights[i];
        DLight& o = dl[i];
        o.kind = l.kind;
        o.flags = (l.two_sided ? 1 : 0) | (l.spectrum_kind == HK_SPEC_ILLUMINANT ? 2 : 0);
        o.scale = l.scale;
        o.area = l.area;
        o.Le_tex = -1;
        float cf[4] = {0, 0, 0, 0};
        if (l.kind == HK_LIGHT_DIFFUSE_AREA) {
            o.Le_tex = l.Le.tex;
            std::memcpy(o.Le_rgba, l.Le.c, 16);
            if (l.Le.tex < 0) bake_bounded(c->r2s_host, l.Le.c[0] * l.scale, l.Le.c[1] * l.scale, l.Le.c[2] * l.scale, cf);  // uplift_rgb(Le*scale): Q3
            std::memcpy(o.v, l.v, sizeof o.v);
            std::memcpy(o.normal, l.normal, sizeof o.normal);
            std::memcpy(o.uv, l.uv, sizeof o.uv);
        } else {
            if (l.spectrum_kind == HK_SPEC_ILLUMINANT) {
                cf[0] = l.poly[0];
                cf[1] = l.poly[1];
                cf[2] = l.poly[2];
                cf[3] = l.illum_scale;
            } else
                bake_illuminant(c->r2s_host, l.i_rgb[0], l.i_rgb[1], l.i_rgb[2], cf);
            if (l.kind == HK_LIGHT_POINT || l.kind == HK_LIGHT_SPOT) std::memcpy(o.p, l.position, 12);
            if (l.kind == HK_LIGHT_DIRECTIONAL || l.kind == HK_LIGHT_SUN) std::memcpy(o.p, l.direction, 12);
            if (l.kind == HK_LIGHT_SPOT) {
                const float* m = l.world_to_light;
                float rows[9] = {m[0], m[1], m[2], m[4], m[5], m[6], m[8], m[9], m[10]};
                std::memcpy(o.v, rows, sizeof rows);
                o.cos_total_width = l.cos_total_width;
                o.cos_falloff_start = l.cos_falloff_start;
            }
            if (l.kind == HK_LIGHT_AMBIENT || l.kind == HK_LIGHT_ENVIRONMENT) has_escape = 1;
            if (l.kind == HK_LIGHT_DIFFUSE_AREA && o.Le_tex >= 0) textured_emitters = 1;
            if (l.kind == HK_LIGHT_ENVIRONMENT) {  // scale::RGBSpectrum rides in Le_rgba, the map index in Le_tex
                std::memcpy(o.Le_rgba, l.i_rgb, 16);
                o.Le_tex = l.envmap;
            }
        }
        o.coef = make_float4(cf[0], cf[1], cf[2], cf[3]);
    }
    HIP_TRY(s->lights.upload(dl.data(), dl.size() * sizeof(DLight)));
    hk::build_light_bvh(d->lights, d->n_lights, s->lbvh);
    static_assert(sizeof(DLightNode) == 64, "light node layout");
    {
        std::vector<DLightNode> tmp(s->lbvh.nodes.size() ? s->lbvh.nodes.size() : 1);
        std::memset(tmp.data(), 0, tmp.size() * sizeof(DLightNode));
        for (size_t i = 0; i < s->lbvh.nodes.size(); ++i) {
            const hk::LightBVHNodeH& n = s->lbvh.nodes[i];
            DLightNode& o = tmp[i];
            // volatile: every intermediate is rounded to binary32 exactly where the device code rounded it
            volatile float c[3], dg[3], r[3];
            for (int k = 0; k < 3; ++k) {
                volatile float sum = n.bmin[k] + n.bmax[k];
                c[k] = sum * 0.5f;
                dg[k] = n.bmax[k] - n.bmin[k];
            }
            for (int k = 0; k < 3; ++k) r[k] = n.bmax[k] - c[k];
            auto dot3 = [](volatile float* a) {
                volatile float xx = a[0] * a[0], yy = a[1] * a[1], zz = a[2] * a[2];
                volatile float xy = xx + yy;
                volatile float t = xy + zz;
                return (float)t;
            };
            for (int k = 0; k < 3; ++k) o.centre[k] = c[k];
            volatile float nd = std::sqrt(dot3(dg));
            o.half_diag = nd * 0.5f;
            o.r2 = dot3(r);
            for (int k = 0; k < 3; ++k) o.w[k] = n.w[k];
            o.phi = n.phi, o.cos_o = n.cos_o, o.cos_e = n.cos_e;
            volatile float cc = n.cos_o * n.cos_o;
            volatile float om = 1.0f - cc;
            o.sin_o = std::sqrt(om > 0.0f ? (float)om : 0.0f);
            o.bits = n.bits;
            o.child1_or_light = n.child1_or_light;
        }
        // SIBLING-PAIR ORDER on the device.  The host tree is in the reference's order (bvh-light-sampler.jl:26-46: child 0 = index + 1,
        // child 1 stored), in which the two children a descent evaluates at every level lie in two unrelated 64-B lines.  The device
        // walks entry 0 = root, entry 1 = unused, then the children of every inner node as ONE 128-B aligned pair (2k, 2k + 1), pairs in
        // breadth-first order (the top of the tree is contiguous); an inner node's `child1_or_light` is the entry of its child 0.  Same
        // nodes, same arithmetic per node, one line per level instead of two; hk_scene_light_bvh_copy still hands out the host order.
        if (!s->lbvh.nodes.empty()) {
            std::vector<DLightNode> pairs(2);
            std::memset(pairs.data(), 0, 2 * sizeof(DLightNode));
            pairs[0] = tmp[0];
            std::vector<std::pair<uint32_t, uint32_t>> todo{{0u, 0u}};   // (host index, device entry) of inner nodes whose children are not placed yet
            for (size_t q = 0; q < todo.size(); ++q) {
                const uint32_t hi = todo[q].first, de = todo[q].second;
                if (tmp[hi].bits & 2u) continue;
                const uint32_t h0 = hi + 1, h1 = tmp[hi].child1_or_light - 1, base = (uint32_t)pairs.size();
                pairs.push_back(tmp[h0]);
                pairs.push_back(tmp[h1]);
                pairs[de].child1_or_light = base;
                todo.emplace_back(h0, base);
                todo.emplace_back(h1, base + 1);
            }
            tmp.swap(pairs);
        }
        HIP_TRY(s->lnodes.upload(tmp.data(), tmp.size() * sizeof(DLightNode)));
        std::vector<uint32_t> tr = s->lbvh.bit_trails;
        if (tr.empty()) tr.resize(1);
        HIP_TRY(s->trails.upload(tr.data(), tr.size() * 4));
        std::vector<int32_t> inf = s->lbvh.infinite;
        if (inf.empty()) inf.resize(1);
        HIP_TRY(s->infinite.upload(inf.data(), inf.size() * 4));
    }
    // ---- environment maps: texels + Distribution2D tables, layouts as the reference holds them ----
    {
        std::vector<DEnvMap> de(d->n_envmaps > 0 ? d->n_envmaps : 1);
        std::memset(de.data(), 0, de.size() * sizeof(DEnvMap));
        for (int i = 0; i < d->n_envmaps; ++i) {
            const hk_envmap& e = d->envmaps[i];
            if (e.width <= 0 || e.height <= 0 || e.nu <= 0 || e.nv <= 0 || !e.data || !e.conditional_func || !e.conditional_cdf || !e.conditional_func_int ||
                !e.marginal_func || !e.marginal_cdf) {
                return fail(HK_ERR_INVALID, "incomplete hk_envmap record");
            }
            auto up = [&](const float* src, size_t n, const float** dst) -> hipError_t {
                DevBuf* b = new DevBuf();
                s->env_data.push_back(b);
                hipError_t err = b->upload(src, n * 4);
                *dst = b->as<float>();
                return err;
            };
            DEnvMap& o = de[i];
            const float* texels = nullptr;
            HIP_TRY(up(e.data, (size_t)e.width * e.height * 4, &texels));
            o.data = reinterpret_cast<const float4*>(texels);
            HIP_TRY(up(e.conditional_func, (size_t)e.nu * e.nv, &o.cond_func));
            HIP_TRY(up(e.conditional_cdf, (size_t)(e.nu + 1) * e.nv, &o.cond_cdf));
            HIP_TRY(up(e.conditional_func_int, (size_t)e.nv, &o.cond_func_int));
            HIP_TRY(up(e.marginal_func, (size_t)e.nv, &o.marg_func));
            HIP_TRY(up(e.marginal_cdf, (size_t)e.nv + 1, &o.marg_cdf));
            o.marg_func_int = e.marginal_func_int;
            o.width = e.width, o.height = e.height, o.nu = e.nu, o.nv = e.nv;
            std::memcpy(o.rot, e.rotation, sizeof o.rot);
        }
        HIP_TRY(s->envmaps.upload(de.data(), de.size() * sizeof(DEnvMap)));
    }
    // ---- media: bake sigma_a / sigma_s / Le with uplift_rgb_unbounded, upload grids / NanoVDB bytes ----
    std::vector<DMedium> dmed(d->n_media > 0 ? d->n_media : 1);
    std::memset(dmed.data(), 0, dmed.size() * sizeof(DMedium));
    for (int i = 0; i < d->n_media; ++i) {
        const hk_medium& m = d->media[i];
        DMedium& o = dmed[i];
        if (m.kind < HK_MEDIUM_HOMOGENEOUS || m.kind > HK_MEDIUM_NANOVDB) {
            return fail(HK_ERR_INVALID, "unknown medium kind");
        }
        if (m.kind == HK_MEDIUM_RGB_GRID && ((!m.sigma_a_grid && !m.sigma_s_grid) || (m.Le_grid && !m.sigma_a_grid))) {  // media.jl:1073-1079
            return fail(HK_ERR_INVALID, "RGBGridMedium needs sigma_a_grid or sigma_s_grid (and sigma_a_grid when Le_grid is given)");
        }
        o.kind = m.kind;
        o.g = m.g;
        float cf[4];
        bake_unbounded(c->r2s_host, m.sigma_a[0], m.sigma_a[1], m.sigma_a[2], cf);
        o.sigma_a = make_float4(cf[0], cf[1], cf[2], cf[3]);
        bake_unbounded(c->r2s_host, m.sigma_s[0], m.sigma_s[1], m.sigma_s[2], cf);
        o.sigma_s = make_float4(cf[0], cf[1], cf[2], cf[3]);
        bake_unbounded(c->r2s_host, m.Le[0], m.Le[1], m.Le[2], cf);
        o.Le = make_float4(cf[0], cf[1], cf[2], cf[3]);
        std::memcpy(o.bmin, m.bounds_min, 12);
        std::memcpy(o.bmax, m.bounds_max, 12);
        std::memcpy(o.r2m, m.render_to_medium, 48);
        std::memcpy(o.res, m.res, 12);
        std::memcpy(o.mres, m.majorant_res, 12);
        std::memcpy(o.inv_mat, m.inv_mat, 36);
        std::memcpy(o.vec, m.vec, 12);
        o.root_off = m.root_offset_1based;
        o.root_table_size = m.root_table_size;
        if (m.kind != HK_MEDIUM_HOMOGENEOUS) {
            if (!m.majorant) {
                return fail(HK_ERR_INVALID, "heterogeneous medium without a majorant grid");
            }
            DevBuf* mb = new DevBuf();
            s->media_data.push_back(mb);
            HIP_TRY(mb->upload(m.majorant, (size_t)m.majorant_res[0] * m.majorant_res[1] * m.majorant_res[2] * 4));
            o.majorant = mb->as<float>();
            // one bit per majorant cell: value exactly 0 (DMedium::maj_zero, majorant_skip_zero)
            const size_t ncell = (size_t)m.majorant_res[0] * m.majorant_res[1] * m.majorant_res[2];
            std::vector<uint32_t> zero((ncell + 31) / 32, 0u);
            for (size_t i = 0; i < ncell; ++i)
                if (m.majorant[i] == 0.0f) zero[i >> 5] |= 1u << (i & 31);
            DevBuf* zb = new DevBuf();
            s->media_data.push_back(zb);
            HIP_TRY(zb->upload(zero.data(), zero.size() * 4));
            o.maj_zero = zb->as<uint32_t>();
        }
        if (m.kind == HK_MEDIUM_GRID) {
            DevBuf* db = new DevBuf();
            s->media_data.push_back(db);
            HIP_TRY(db->upload(m.density, (size_t)m.res[0] * m.res[1] * m.res[2] * 4));
            o.density = db->as<float>();
        }
        if (m.kind == HK_MEDIUM_RGB_GRID) {
            const size_t nvox = (size_t)m.res[0] * m.res[1] * m.res[2];
            const float* src[3] = {m.sigma_a_grid, m.sigma_s_grid, m.Le_grid};
            const float4** dst[3] = {&o.rgb_a, &o.rgb_s, &o.rgb_Le};
            for (int g = 0; g < 3; ++g) {
                if (!src[g]) continue;
                DevBuf* gb = new DevBuf();
                s->media_data.push_back(gb);
                HIP_TRY(gb->upload(src[g], nvox * 16));
                *dst[g] = gb->as<float4>();
            }
            o.sigma_scale = m.sigma_scale;
            o.Le_scale = m.Le_scale;
        }
        if (m.kind == HK_MEDIUM_NANOVDB) {
            DevBuf* nb = new DevBuf();
            s->media_data.push_back(nb);
            HIP_TRY(nb->upload(m.nvdb_bytes, (size_t)m.nvdb_size));
            o.nvdb = nb->as<unsigned char>();
            // flatten the tree into the block table.  Extent = every 8^3 block that is a leaf or lies in a non-background tile (the
            // node tables are scanned, so nothing the walk could return differently is left outside), united with the record's
            // index bbox, +1 block of margin for the trilinear +1 taps.  Blocks further out hold the root background.
            float bg_probe;
            {
                const unsigned char* B = m.nvdb_bytes;
                bg_probe = hknv::f32(B, m.root_offset_1based + 28);
                int lo[3] = {m.index_bbox_min[0] >> 3, m.index_bbox_min[1] >> 3, m.index_bbox_min[2] >> 3};
                int hi[3] = {m.index_bbox_max[0] >> 3, m.index_bbox_max[1] >> 3, m.index_bbox_max[2] >> 3};
                auto include = [&](long long x0, long long y0, long long z0, long long nblk) {  // voxel origin, extent in blocks
                    long long o[3] = {x0 >> 3, y0 >> 3, z0 >> 3};
                    for (int k = 0; k < 3; ++k) {
                        if (o[k] < lo[k]) lo[k] = (int)o[k];
                        if (o[k] + nblk - 1 > hi[k]) hi[k] = (int)(o[k] + nblk - 1);
                    }
                };
                auto sext21 = [](unsigned long long v) { return (long long)((v & 0x100000ull) ? (v | ~0x1fffffull) : v); };
                bool unbounded = false;
                for (int i = 0; i < m.root_table_size; ++i) {
                    long long tile = m.root_offset_1based + 64 + (long long)i * 32;
                    unsigned long long key = (unsigned long long)hknv::i64(B, tile);
                    long long tz = sext21(key & 0x1fffffull) << 12, ty = sext21((key >> 21) & 0x1fffffull) << 12, tx = sext21((key >> 42) & 0x1fffffull) << 12;
                    long long child = hknv::i64(B, tile + 8);
                    if (child == 0) {
                        if (hknv::f32(B, tile + 20) != bg_probe) unbounded = true;  // a 4096^3 constant tile: not tabulated
                        continue;
                    }
                    long long upper = m.root_offset_1based + child;
                    for (int nu = 0; nu < 32768; ++nu) {
                        long long ux = tx + (((nu >> 10) & 31) << 7), uy = ty + (((nu >> 5) & 31) << 7), uz = tz + ((nu & 31) << 7);
                        if (!hknv::mask(B, upper + 4128, nu)) {
                            if (hknv::f32(B, upper + 8256 + (long long)nu * 8) != bg_probe) include(ux, uy, uz, 16);
                            continue;
                        }
                        long long lower = upper + hknv::i64(B, upper + 8256 + (long long)nu * 8);
                        for (int nl = 0; nl < 4096; ++nl) {
                            long long lx = ux + (((nl >> 8) & 15) << 3), ly = uy + (((nl >> 4) & 15) << 3), lz = uz + ((nl & 15) << 3);
                            if (hknv::mask(B, lower + 544, nl) || hknv::f32(B, lower + 1088 + (long long)nl * 8) != bg_probe) include(lx, ly, lz, 1);
                        }
                    }
                }
                if (unbounded) {
                    return fail(HK_ERR_UNSUPPORTED, "NanoVDB grid with a non-background root tile (4096^3 constant region) is not supported");
                }
                for (int k = 0; k < 3; ++k) {
                    o.nvb_min[k] = lo[k] - 1;
                    o.nvb_dim[k] = hi[k] - lo[k] + 3;
                }
            }
            long long dim[3], total = 1;
            for (int k = 0; k < 3; ++k) {
                dim[k] = o.nvb_dim[k] < 3 ? 3 : o.nvb_dim[k];
                total *= dim[k];
            }
            if (total > (1ll << 27) || m.nvdb_size >= (1ll << 32)) {
                return fail(HK_ERR_UNSUPPORTED, "NanoVDB grid too large for the device block table (index bbox > 2^27 blocks or buffer >= 4 GiB)");
            }
            o.nv_background = bg_probe;
            std::vector<uint2> table((size_t)total);
            bool margin_ok = true;
            for (long long bx = 0; bx < dim[0]; ++bx)
                for (long long by = 0; by < dim[1]; ++by)
                    for (long long bz = 0; bz < dim[2]; ++bz) {
                        float value;
                        long long leaf = hknv::find_block(m.nvdb_bytes, m.root_offset_1based, m.root_table_size, (int)((o.nvb_min[0] + bx) * 8), (int)((o.nvb_min[1] + by) * 8),
                                                          (int)((o.nvb_min[2] + bz) * 8), value);
                        uint32_t bits;
                        std::memcpy(&bits, &value, 4);
                        table[(size_t)bz + (size_t)dim[2] * ((size_t)by + (size_t)dim[1] * (size_t)bx)] = make_uint2((uint32_t)leaf, bits);
                        const bool on_margin = bx == 0 || by == 0 || bz == 0 || bx == dim[0] - 1 || by == dim[1] - 1 || bz == dim[2] - 1;
                        if (on_margin && (leaf != 0 || value != bg_probe)) margin_ok = false;  // cannot happen after the scan above
                    }
            if (!margin_ok) {
                return fail(HK_ERR_INVALID, "NanoVDB block table: non-background data on the margin (corrupt tree?)");
            }
            DevBuf* tb = new DevBuf();
            s->media_data.push_back(tb);
            HIP_TRY(tb->upload(table.data(), table.size() * sizeof(uint2)));
            o.nv_blocks = tb->as<uint2>();
            for (int k = 0; k < 3; ++k) o.nvb_dim[k] = (int)dim[k];
            // dense bricks: every block of the table materialised (leaf values copied, constant blocks filled) WITH the first voxel
            // plane of its +x / +y / +z neighbours (9^3 floats per block, z fastest), so that the device fetches the eight taps of a
            // lookup from one brick at one computed address.  2.9 KB per block: taken when it fits HK_NVDB_DENSE_MB (default 4096 MB —
            // 288 GB of HBM are there to be used; the bench cloud needs 67 MB), else the table + leaves path stays.
            {
                size_t budget_mb = 4096;
                if (const char* e = hk::knob("HK_NVDB_DENSE_MB")) budget_mb = (size_t)std::atol(e);
                const size_t brick_bytes = (size_t)total * 729 * sizeof(float);
                if (brick_bytes <= budget_mb * (size_t)(1 << 20)) {
                    std::vector<float> plain((size_t)total * 512);
                    for (size_t b = 0; b < (size_t)total; ++b) {
                        const uint2 e = table[b];
                        float* dst = plain.data() + b * 512;
                        if (e.x == 0u) {
                            float v;
                            std::memcpy(&v, &e.y, 4);
                            for (int n = 0; n < 512; ++n) dst[n] = v;
                        } else
                            for (int n = 0; n < 512; ++n) dst[n] = hknv::f32(m.nvdb_bytes, (long long)e.x + 96 + (long long)n * 4);
                    }
                    auto voxel = [&](long long bx, long long by, long long bz, int x, int y, int z) -> float {   // (x, y, z) in 0 .. 8 relative to block (bx, by, bz)
                        bx += x >> 3, by += y >> 3, bz += z >> 3;
                        if (bx >= dim[0] || by >= dim[1] || bz >= dim[2]) return bg_probe;   // beyond the (background) margin
                        return plain[((size_t)bz + (size_t)dim[2] * ((size_t)by + (size_t)dim[1] * (size_t)bx)) * 512 + (size_t)(((x & 7) << 6) | ((y & 7) << 3) | (z & 7))];
                    };
                    std::vector<float> bricks((size_t)total * 729);
                    for (long long bx = 0; bx < dim[0]; ++bx)
                        for (long long by = 0; by < dim[1]; ++by)
                            for (long long bz = 0; bz < dim[2]; ++bz) {
                                float* dst = bricks.data() + ((size_t)bz + (size_t)dim[2] * ((size_t)by + (size_t)dim[1] * (size_t)bx)) * 729;
                                for (int x = 0; x < 9; ++x)
                                    for (int y = 0; y < 9; ++y)
                                        for (int z = 0; z < 9; ++z) dst[x * 81 + y * 9 + z] = voxel(bx, by, bz, x, y, z);
                            }
                    DevBuf* bb = new DevBuf();
                    s->media_data.push_back(bb);
                    HIP_TRY(bb->upload(bricks.data(), brick_bytes));
                    o.nv_bricks = bb->as<float>();
                }
            }
        }
    }
    HIP_TRY(s->media.upload(dmed.data(), dmed.size() * sizeof(DMedium)));
    DScene& D = s->d;
    D.media = s->media.as<DMedium>();
    D.n_media = d->n_media;
    D.media_mask = 0;
    for (int i = 0; i < d->n_media; ++i) D.media_mask |= 1 << d->media[i].kind;
    D.all_grey = d->n_media == 1 ? 1 : 0;   // the GREY kernels read ONE medium record
    for (int i = 0; i < d->n_media; ++i) {
        const DMedium& dm = dmed[i];
        const bool flat_a = dm.sigma_a.w == 0.0f || (dm.sigma_a.x == 0.0f && dm.sigma_a.y == 0.0f);
        const bool flat_s = dm.sigma_s.w == 0.0f || (dm.sigma_s.x == 0.0f && dm.sigma_s.y == 0.0f);
        if (!(flat_a && flat_s && (dm.kind == HK_MEDIUM_GRID || dm.kind == HK_MEDIUM_NANOVDB))) D.all_grey = 0;
    }
    D.grey_pool = (D.all_grey && D.n_media == 1 && dmed[0].mres[0] <= 1024 && dmed[0].mres[1] <= 1024 && dmed[0].mres[2] <= 1024) ? 1 : 0;
    D.grey_bricks = (D.all_grey && dmed[0].kind == HK_MEDIUM_NANOVDB && dmed[0].nv_bricks != nullptr) ? 1 : 0;
    D.nodes = s->nodes.as<DNode>();
    D.qnodes = have_qnodes ? s->qnodes.as<DQNode>() : nullptr;
    for (int k = 0; k < 3; ++k) D.q_base[k] = q_base[k], D.q_cell[k] = q_cell[k];
    D.leaf_tris = s->leaf_tris.as<float4>();
    D.root_ref = bvh.root_ref;
    D.n_tris = T;
    D.n_nodes = (int)bvh.nodes.size();
    D.pad_nodes = 0;
    D.positions = s->positions.as<float>();
    D.normals = d->normals ? s->normals.as<float>() : nullptr;
    D.uvs = d->uvs ? s->uvs.as<float>() : nullptr;
    D.tangents = d->tangents ? s->tangents.as<float>() : nullptr;
    D.meta = s->meta.as<DTriMeta>();
    D.tri_shade = s->tri_shade.p ? s->tri_shade.as<float>() : nullptr;
    D.materials = s->materials.as<DMaterial>();
    D.textures = s->textures.as<DTexture>();
    D.spectra = s->spectra.as<DPLSpectrum>();
    D.mis = s->mis.as<DMediumInterface>();
    D.lights = s->lights.as<DLight>();
    D.n_lights = d->n_lights;
    D.n_materials = d->n_materials;
    D.lnodes = s->lnodes.as<DLightNode>();
    D.bit_trails = s->trails.as<uint32_t>();
    D.infinite_lights = s->infinite.as<int>();
    D.num_bvh_lights = s->lbvh.num_bvh;
    D.num_infinite_lights = (int)s->lbvh.infinite.size();
    D.envmaps = s->envmaps.as<DEnvMap>();
    D.n_envmaps = d->n_envmaps;
    D.has_escape_lights = has_escape;
    D.simple_lights = (!has_escape && !textured_emitters && d->n_textures == 0) ? 1 : 0;
    D.all_opaque = all_opaque ? 1 : 0;
    D.bvh_depth = bvh.max_depth;
    *out = guard.release();
    return HK_OK;
}
extern "C" int32_t hk_scene_destroy(hk_scene* s) {
    int status = HK_OK;   // the object goes either way; a device error of a pass that was still noted / in flight is reported (hk_last_error)
    if (s) {
        (void)hipSetDevice(s->ctx->device);
        status = join_lanes(s->ctx);
        if (hipStreamSynchronize(s->ctx->stream) != hipSuccess && status == HK_OK) status = fail(HK_ERR_DEVICE, "hk_scene_destroy: the context's stream reports an error");
        delete s;
    }
    return status;
}
extern "C" int32_t hk_scene_bvh_info(hk_scene* s, int32_t* n_nodes, int32_t* n_leaf_tris, int32_t* max_depth) {
    if (!s) return fail(HK_ERR_INVALID, "null scene");
    if (n_nodes) *n_nodes = s->bvh_nodes;
    if (n_leaf_tris) *n_leaf_tris = s->bvh_leaf_tris;
    if (max_depth) *max_depth = s->bvh_depth;
    return HK_OK;
}
extern "C" int32_t hk_scene_light_bvh_copy(hk_scene* s, int32_t* n_nodes, float* nodes_out, uint32_t* bit_trails) {
    if (!s || !n_nodes) return fail(HK_ERR_INVALID, "null argument");
    *n_nodes = (int32_t)s->lbvh.nodes.size();
    if (nodes_out)
        for (size_t i = 0; i < s->lbvh.nodes.size(); ++i) {
            const hk::LightBVHNodeH& n = s->lbvh.nodes[i];
            float* o = nodes_out + 16 * i;
            std::memcpy(o, n.bmin, 12);
            std::memcpy(o + 3, n.bmax, 12);
            std::memcpy(o + 6, n.w, 12);
            o[9] = n.phi;
            o[10] = n.cos_o;
            o[11] = n.cos_e;
            o[12] = (n.bits & 1u) ? 1.0f : 0.0f;
            o[13] = (float)n.child1_or_light;
            o[14] = (n.bits & 2u) ? 1.0f : 0.0f;
            o[15] = 0.0f;
        }
    if (bit_trails)
        for (size_t i = 0; i < s->lbvh.bit_trails.size(); ++i) bit_trails[i] = s->lbvh.bit_trails[i];
    return HK_OK;
}

// ---- integrator: filter tabulation (GPUFilterSamplerData, filter.jl:636-725) -------------------------
namespace {
float gaussian_1d(float x, float sigma) { return std::exp(-(x * x) / (2.0f * (sigma * sigma))); }
float mitchell_1d(float x, float B, float C) {
    x = std::fabs(x);
    if (x <= 1.0f) return ((12.0f - 9.0f * B - 6.0f * C) * (x * x * x) + (-18.0f + 12.0f * B + 6.0f * C) * (x * x) + (6.0f - 2.0f * B)) / 6.0f;
    if (x <= 2.0f) return ((-B - 6.0f * C) * (x * x * x) + (6.0f * B + 30.0f * C) * (x * x) + (-12.0f * B - 48.0f * C) * x + (8.0f * B + 24.0f * C)) / 6.0f;
    return 0.0f;
}
float sinc1(float x) {
    x = std::fabs(x);
    if (x < 1e-5f) return 1.0f;
    x *= 3.14159265358979323846f;
    return std::sin(x) / x;
}
float wsinc(float x, float r, float tau) {
    x = std::fabs(x);
    if (x > r) return 0.0f;
    return sinc1(x) * sinc1(x / tau);
}
float filter_eval(const hk_integrator_params& p, float ex, float ey, float x, float y) {
    float rx = p.filter_radius[0], ry = p.filter_radius[1];
    switch (p.filter_type) {
        case HK_FILTER_BOX: return (std::fabs(x) <= rx && std::fabs(y) <= ry) ? 1.0f : 0.0f;
        case HK_FILTER_TRIANGLE: return std::fmax(0.0f, rx - std::fabs(x)) * std::fmax(0.0f, ry - std::fabs(y));
        case HK_FILTER_GAUSSIAN: return std::fmax(0.0f, gaussian_1d(x, p.filter_param1) - ex) * std::fmax(0.0f, gaussian_1d(y, p.filter_param1) - ey);
        case HK_FILTER_MITCHELL: return mitchell_1d(2.0f * x / rx, p.filter_param1, p.filter_param2) * mitchell_1d(2.0f * y / ry, p.filter_param1, p.filter_param2);
        case HK_FILTER_LANCZOS: return wsinc(x, rx, p.filter_param1) * wsinc(y, ry, p.filter_param1);
    }
    return 0.0f;
}
}  // namespace

extern "C" int32_t hk_integrator_create(hk_ctx* c, const hk_integrator_params* p, hk_integrator** out) {
    if (!c || !p || !out) return fail(HK_ERR_INVALID, "null argument");
    if (p->max_depth < 1 || p->max_depth > 255) return fail(HK_ERR_INVALID, "max_depth must be in 1..255");
    if (p->filter_type < HK_FILTER_BOX || p->filter_type > HK_FILTER_LANCZOS) return fail(HK_ERR_INVALID, "bad filter type");
    if (p->material_coherence < 0 || p->material_coherence > 2) return fail(HK_ERR_INVALID, "material_coherence must be :none, :sorted, :per_type");
    HIP_TRY(hipSetDevice(c->device));
    hk_integrator* I = new hk_integrator();
    I->ctx = c;
    I->p = *p;
    DFilter& f = I->filter;
    f.type = p->filter_type;
    f.rx = p->filter_radius[0];
    f.ry = p->filter_radius[1];
    f.p1 = p->filter_param1;
    f.p2 = p->filter_param2;
    if (f.type != HK_FILTER_BOX && f.type != HK_FILTER_TRIANGLE) {
        float ex = 0, ey = 0;
        if (f.type == HK_FILTER_GAUSSIAN) {
            ex = gaussian_1d(f.rx, f.p1);
            ey = gaussian_1d(f.ry, f.p1);
        }
        int nx = (int)std::ceil(32 * f.rx), ny = (int)std::ceil(32 * f.ry);
        if (nx < 8) nx = 8;
        if (ny < 8) ny = 8;
        float dmin_x = -f.rx, dmin_y = -f.ry, dmax_x = f.rx, dmax_y = f.ry;
        float dx = (dmax_x - dmin_x) / (float)nx, dy = (dmax_y - dmin_y) / (float)ny;
        std::vector<float> func((size_t)nx * ny), mfunc(ny, 0.0f), mcdf(ny + 1, 0.0f), ccdf((size_t)ny * (nx + 1), 0.0f);
        for (int iy = 1; iy <= ny; ++iy)
            for (int ix = 1; ix <= nx; ++ix) {
                float px = dmin_x + ((float)ix - 0.5f) * dx, py = dmin_y + ((float)iy - 0.5f) * dy;
                float v = filter_eval(*p, ex, ey, px, py);
                func[(size_t)(iy - 1) * nx + ix - 1] = v > 0.0f ? v : 0.0f;
            }
        for (int iy = 0; iy < ny; ++iy)
            for (int ix = 0; ix < nx; ++ix) mfunc[iy] += func[(size_t)iy * nx + ix];
        for (int iy = 0; iy < ny; ++iy) mcdf[iy + 1] = mcdf[iy] + mfunc[iy];
        float func_integral = mcdf[ny] * dx * dy;
        float end = mcdf[ny];
        if (end > 0.0f)
            for (auto& v : mcdf) v /= end;
        else
            for (int iy = 0; iy <= ny; ++iy) mcdf[iy] = (float)iy / (float)ny;
        for (int iy = 0; iy < ny; ++iy) {
            float* row = &ccdf[(size_t)iy * (nx + 1)];
            for (int ix = 0; ix < nx; ++ix) row[ix + 1] = row[ix] + func[(size_t)iy * nx + ix];
            float rs = row[nx];
            if (rs > 0.0f)
                for (int ix = 0; ix <= nx; ++ix) row[ix] /= rs;
            else
                for (int ix = 0; ix <= nx; ++ix) row[ix] = (float)ix / (float)nx;
        }
        HIP_TRY(I->f_func.upload(func.data(), func.size() * 4));
        HIP_TRY(I->f_mcdf.upload(mcdf.data(), mcdf.size() * 4));
        HIP_TRY(I->f_mfunc.upload(mfunc.data(), mfunc.size() * 4));
        HIP_TRY(I->f_ccdf.upload(ccdf.data(), ccdf.size() * 4));
        f.nx = nx;
        f.ny = ny;
        f.func = I->f_func.as<float>();
        f.marginal_cdf = I->f_mcdf.as<float>();
        f.marginal_func = I->f_mfunc.as<float>();
        f.conditional_cdf = I->f_ccdf.as<float>();
        f.dmin_x = dmin_x;
        f.dmin_y = dmin_y;
        f.dmax_x = dmax_x;
        f.dmax_y = dmax_y;
        f.func_integral = func_integral;
    }
    *out = I;
    return HK_OK;
}
extern "C" int32_t hk_integrator_destroy(hk_integrator* I) {
    int status = HK_OK;   // the object goes either way; a device error of a pass that was still noted / in flight is reported (hk_last_error)
    if (I) {
        (void)hipSetDevice(I->ctx->device);
        status = join_lanes(I->ctx);
        if (hipStreamSynchronize(I->ctx->stream) != hipSuccess && status == HK_OK) status = fail(HK_ERR_DEVICE, "hk_integrator_destroy: the context's stream reports an error");
        quiesce(I->ctx);
        delete I;
    }
    return status;
}

// ---- film ---------------------------------------------------------------------------------------------
extern "C" int32_t hk_film_create(hk_ctx* c, int32_t w, int32_t h, int32_t f64, void* external, hk_film** out) {
    if (!c || !out || w < 1 || h < 1) return fail(HK_ERR_INVALID, "bad film arguments");
    HIP_TRY(hipSetDevice(c->device));
    hk_film* f = new hk_film();
    f->ctx = c;
    f->width = w;
    f->height = h;
    f->f64 = f64 != 0;
    size_t bytes = (size_t)4 * w * h * (f64 ? 8 : 4);
    f->external = external != nullptr;
    if (external)
        f->accum = external;
    else {
        HIP_TRY(f->own.alloc(bytes));
        f->accum = f->own.p;
        HIP_TRY(hipMemsetAsync(f->accum, 0, bytes, c->stream));
    }
    *out = f;
    return HK_OK;
}
extern "C" int32_t hk_film_destroy(hk_film* f) {
    int status = HK_OK;   // the object goes either way; a device error of a pass that was still noted / in flight is reported (hk_last_error)
    if (f) {
        (void)hipSetDevice(f->ctx->device);
        status = join_lanes(f->ctx);
        if (hipStreamSynchronize(f->ctx->stream) != hipSuccess && status == HK_OK) status = fail(HK_ERR_DEVICE, "hk_film_destroy: the context's stream reports an error");
        if (f->pinned_user) (void)hipHostUnregister(f->pinned_user);
        for (auto& st : f->staging)
            if (st) (void)hipHostFree(st);
        if (f->ev_read) (void)hipEventDestroy(f->ev_read);
        delete f;
    }
    return status;
}
extern "C" int32_t hk_film_clear(hk_film* f) {
    if (!f) return fail(HK_ERR_INVALID, "null film");
    HIP_TRY(hipSetDevice(f->ctx->device));
    if (int e = join_lanes(f->ctx)) return e;
    HIP_TRY(hipMemsetAsync(f->accum, 0, (size_t)4 * f->width * f->height * (f->f64 ? 8 : 4), f->ctx->stream));
    return HK_OK;
}
extern "C" void* hk_film_accum_device_ptr(hk_film* f) {
    if (f) {
        (void)flush_pending(f->ctx);   // (whoever asks for the pointer is about to look: the noted calls are enqueued first)
        f->exposed = true;             // ... and may keep it: from now on calls into this film are enqueued at once (hk_render: ORDERING CONTRACT)
    }
    return f ? f->accum : nullptr;
}
extern "C" int32_t hk_film_read_accum(hk_ctx* c, hk_film* f, void* out) {
    if (!c || !f || !out) return fail(HK_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(c->device));
    if (int e = join_lanes(c)) return e;
    HIP_TRY(hipMemcpyAsync(out, f->accum, (size_t)4 * f->width * f->height * (f->f64 ? 8 : 4), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return HK_OK;
}
// K13 + the device-to-host copy of the frame.  A pageable destination made this the most expensive part of an interactive viewer's loop
// (round 4: 3.7 ms of a 5.3-ms one-sample call at 800^2 — the runtime stages a pageable copy through small pinned buffers and the
// stream waits for each).  Now the frame lands in PINNED memory: the film's own staging buffers (two, in turn), or the caller's
// buffer itself once the same pointer has come twice in a row and hipHostRegister accepted it (a viewer reads into ONE framebuffer).
static int enqueue_frame_read(hk_ctx* c, hk_film* f, float* direct) {
    const size_t bytes = (size_t)3 * f->width * f->height * 4;
    if (int e = join_lanes(c)) return e;
    if (f->readback.bytes != bytes) HIP_TRY(f->readback.alloc(bytes));
    if (!f->ev_read) HIP_TRY(hipEventCreateWithFlags(&f->ev_read, hipEventDisableTiming));
    float* dst = direct;
    if (!dst) {
        const int k = f->staging_next;
        if (!f->staging[k]) HIP_TRY(hipHostMalloc((void**)&f->staging[k], bytes, hipHostMallocDefault));
        dst = f->staging[k];
        f->staging_last = k;
        f->staging_next = k ^ 1;
    }
    hk::launch_finalize(c->stream, f->accum, f->f64, f->readback.as<float>(), f->width, f->height);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(dst, f->readback.p, bytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipEventRecord(f->ev_read, c->stream));
    return HK_OK;
}
extern "C" int32_t hk_film_read_rgb(hk_ctx* c, hk_film* f, float* out) {
    if (!c || !f || !out) return fail(HK_ERR_INVALID, "null argument");
    if (f->ctx != c) return fail(HK_ERR_INVALID, "film belongs to another context");
    HIP_TRY(hipSetDevice(c->device));
    KnobScope knobs(&c->knobs);
    const size_t bytes = (size_t)3 * f->width * f->height * 4;
    if (f->read_in_flight) {   // an asynchronous read nobody waited for: its staging buffer is simply overtaken
        HIP_TRY(hipEventSynchronize(f->ev_read));
        f->read_in_flight = false;
    }
    // Caller memory is registered with the driver only on request (the library cannot know when a buffer it registered by
    // itself is freed — a C host that mallocs a frame buffer per frame gets the same address back, and the copy would go through a
    // stale registration): hk_film_pin_host names the buffer, or HK_READBACK_PIN=1 brings back "the same pointer twice in a row".
    // A registration that failed is remembered for that pointer and not tried again on every frame.
    const char* pin = hk::knob("HK_READBACK_PIN");
    if (!f->pinned_explicit) {
        if (pin && std::atoi(pin) == 1 && f->pinned_user != out && f->last_out == out && f->pin_failed != out) {
            if (f->pinned_user) (void)hipHostUnregister(f->pinned_user);
            f->pinned_user = hipHostRegister(out, bytes, hipHostRegisterDefault) == hipSuccess ? out : nullptr;
            if (!f->pinned_user) f->pin_failed = out;
            (void)hipGetLastError();
        } else if (f->pinned_user && f->pinned_user != out) {
            (void)hipHostUnregister(f->pinned_user);
            f->pinned_user = nullptr;
        }
    }
    f->last_out = out;
    const bool direct = f->pinned_user == out;
    if (int e = enqueue_frame_read(c, f, direct ? out : nullptr)) return e;
    HIP_TRY(hipEventSynchronize(f->ev_read));
    if (!direct) std::memcpy(out, f->staging[f->staging_last], bytes);
    return HK_OK;
}
extern "C" int32_t hk_film_pin_host(hk_film* f, float* host_hw3) {
    if (!f || !host_hw3) return fail(HK_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(f->ctx->device));
    if (f->read_in_flight) {
        HIP_TRY(hipEventSynchronize(f->ev_read));
        f->read_in_flight = false;
    }
    if (f->pinned_user == host_hw3 && f->pinned_explicit) return HK_OK;
    if (f->pinned_user) (void)hipHostUnregister(f->pinned_user);
    f->pinned_user = nullptr, f->pinned_explicit = false;
    if (hipHostRegister(host_hw3, (size_t)3 * f->width * f->height * 4, hipHostRegisterDefault) != hipSuccess) {
        (void)hipGetLastError();
        return fail(HK_ERR_DEVICE, "hk_film_pin_host: hipHostRegister refused the buffer");
    }
    f->pinned_user = host_hw3, f->pinned_explicit = true;
    return HK_OK;
}
extern "C" int32_t hk_film_unpin_host(hk_film* f) {
    if (!f) return fail(HK_ERR_INVALID, "null film");
    HIP_TRY(hipSetDevice(f->ctx->device));
    if (f->read_in_flight) {
        HIP_TRY(hipEventSynchronize(f->ev_read));
        f->read_in_flight = false;
    }
    if (f->pinned_user) (void)hipHostUnregister(f->pinned_user);
    f->pinned_user = nullptr, f->pinned_explicit = false, f->last_out = nullptr;
    return HK_OK;
}
extern "C" int32_t hk_film_read_rgb_async(hk_ctx* c, hk_film* f) {
    if (!c || !f) return fail(HK_ERR_INVALID, "null argument");
    if (f->ctx != c) return fail(HK_ERR_INVALID, "film belongs to another context");
    HIP_TRY(hipSetDevice(c->device));
    KnobScope knobs(&c->knobs);
    if (f->read_in_flight) HIP_TRY(hipEventSynchronize(f->ev_read));   // (two buffers: the one about to be refilled is the one BEFORE the last)
    if (int e = enqueue_frame_read(c, f, nullptr)) return e;
    f->read_in_flight = true;
    return HK_OK;
}
extern "C" int32_t hk_film_read_wait(hk_ctx* c, hk_film* f, float* out, const float** frame) {
    if (!c || !f) return fail(HK_ERR_INVALID, "null argument");
    if (f->ctx != c) return fail(HK_ERR_INVALID, "film belongs to another context");
    if (f->staging_last < 0 || !f->ev_read) return fail(HK_ERR_INVALID, "hk_film_read_wait without hk_film_read_rgb_async");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipEventSynchronize(f->ev_read));
    f->read_in_flight = false;
    const float* src = f->staging[f->staging_last];
    if (out) std::memcpy(out, src, (size_t)3 * f->width * f->height * 4);
    if (frame) *frame = src;
    return HK_OK;
}

namespace {
int run_postprocess(hk_ctx* c, const hk_postprocess_params* P, int w, int h, const float* src_dev, const float* depth_host, float* dst_host, DevBuf& out) {
    const size_t bytes = (size_t)3 * w * h * 4;
    HIP_TRY(out.alloc(bytes));
    DevBuf dd;
    if (depth_host && P->mask_escaped) HIP_TRY(dd.upload(depth_host, (size_t)w * h * 4));
    hk::launch_postprocess(c->stream, *P, src_dev, dd.as<float>(), out.as<float>(), h, w);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(dst_host, out.p, bytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return HK_OK;
}
}  // namespace
extern "C" int32_t hk_film_postprocess(hk_ctx* c, hk_film* f, const hk_postprocess_params* P, const float* depth, float* dst) {
    if (!c || !f || !P || !dst) return fail(HK_ERR_INVALID, "null argument");
    if (P->tonemap < HK_TONEMAP_NONE || P->tonemap > HK_TONEMAP_FILMIC) return fail(HK_ERR_INVALID, "unknown tonemap");
    HIP_TRY(hipSetDevice(c->device));
    size_t bytes = (size_t)3 * f->width * f->height * 4;
    if (int e = join_lanes(c)) return e;
    if (f->readback.bytes != bytes) HIP_TRY(f->readback.alloc(bytes));
    hk::launch_finalize(c->stream, f->accum, f->f64, f->readback.as<float>(), f->width, f->height);
    DevBuf out;
    return run_postprocess(c, P, f->width, f->height, f->readback.as<float>(), depth, dst, out);
}
extern "C" int32_t hk_postprocess(hk_ctx* c, const hk_postprocess_params* P, int32_t w, int32_t h, const float* src, const float* depth, float* dst) {
    if (!c || !P || !src || !dst || w <= 0 || h <= 0) return fail(HK_ERR_INVALID, "bad argument");
    if (P->tonemap < HK_TONEMAP_NONE || P->tonemap > HK_TONEMAP_FILMIC) return fail(HK_ERR_INVALID, "unknown tonemap");
    HIP_TRY(hipSetDevice(c->device));
    DevBuf in, out;
    HIP_TRY(in.upload(src, (size_t)3 * w * h * 4));
    return run_postprocess(c, P, w, h, in.as<float>(), depth, dst, out);
}

extern "C" int32_t hk_denoise(hk_ctx* c, const hk_denoise_params* P, int32_t w, int32_t h, const float* src, const float* normal, const float* depth, float* dst,
                              float* src_after) {
    if (!c || !P || !src || !normal || !depth || !dst || w <= 0 || h <= 0) return fail(HK_ERR_INVALID, "bad argument");
    if (P->iterations < 0 || P->iterations > 30) return fail(HK_ERR_INVALID, "iterations out of range");
    HIP_TRY(hipSetDevice(c->device));
    const size_t npx = (size_t)w * h;
    DevBuf a, b, nn, dd, var;
    HIP_TRY(a.upload(src, npx * 12));
    HIP_TRY(b.alloc(npx * 12));
    HIP_TRY(nn.upload(normal, npx * 12));
    HIP_TRY(dd.upload(depth, npx * 4));
    HIP_TRY(var.alloc(npx * 4));
    if (P->use_variance) hk::launch_denoise_variance(c->stream, a.as<float>(), var.as<float>(), h, w);
    for (int i = 1; i <= P->iterations; ++i) {   // odd passes a -> b, even passes b -> a (denoise.jl:337-361)
        const float* in = (i & 1) ? a.as<float>() : b.as<float>();
        float* out = (i & 1) ? b.as<float>() : a.as<float>();
        hk::launch_denoise_atrous(c->stream, *P, 1 << (i - 1), in, nn.as<float>(), dd.as<float>(), var.as<float>(), out, h, w);
    }
    HIP_TRY(hipGetLastError());
    // film.postprocess <- the last written buffer; with iterations == 0 that is the untouched framebuffer (:365-371)
    HIP_TRY(hipMemcpyAsync(dst, (P->iterations & 1) ? b.p : a.p, npx * 12, hipMemcpyDeviceToHost, c->stream));
    if (src_after) HIP_TRY(hipMemcpyAsync(src_after, a.p, npx * 12, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return HK_OK;
}

// ---- path state -----------------------------------------------------------------------------------------
namespace {
// the path-state arrays are carved from ONE allocation (a slab: see SlabCache) instead of ~40; HK_STATE_SLAB=0: one allocation each
template <class T>
hipError_t alloc_arr(hk_integrator* I, T*& dst, size_t n) {
    if (I->slab_mode != 0) {
        const size_t bytes = (n * sizeof(T) + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1);
        dst = I->slab_mode == 2 ? reinterpret_cast<T*>(static_cast<char*>(I->slab_base) + I->slab_off) : nullptr;
        I->slab_off += bytes;
        return hipSuccess;
    }
    DevBuf* b = new DevBuf();
    I->bufs.push_back(b);
    hipError_t e = b->alloc(n * sizeof(T));
    dst = b->as<T>();
    return e;
}
// W virtual wave segments: every queue is split W ways and a segment is processed by one wave per kernel, so W is independent of each
// kernel's residency.  Finer segments balance better, but every segment should keep >= 16 chunks (1024 paths) so its per-kind queues
// fill whole waves: W = chunks / 16, capped per CU.  Measured on the four bench scenes (work lists + 64-way XCD-affine tickets):
//   media                     tickets, 96 per CU    (round 3, the 5 %-fill cloud: 1.22 s per frame at 256, 1.14 at 96, 1.18 at 64; the round-2 blob cloud preferred 256)
//   surfaces, closed scene    tickets, 192 per CU   (Cornell 7.19 -> 7.55, many-light 1.11 -> 1.19: trace -10 %, light-BVH shading -9 %)
//   surfaces, open scene      static stride, 48 per CU   (sky: most paths escape after 1-2 vertices; finer segments only fragment
//                             the per-kind queues: k_shade +7 % at 96, +15 % at 256, whichever way they are handed out)
// HK_WAVES_PER_CU / HK_DYNAMIC_SEGMENTS override.  stats rows are indexed by PHYSICAL wave (ctx->stat_rows).
int ensure_state(hk_integrator* I, int capacity, bool media, bool open_scene, bool small_scene, bool grey_compact, hipStream_t users = nullptr, bool fusable = false, bool opaque_surfaces = false) {
    const int n_cu = I->ctx->n_cu;
    // a closed scene's pass of at most HK_MID_PASS_PATHS_M (48) million paths — a rank's share of a frame under 8-way strong scaling —
    // behaves like the open scene WHEN THE SCENE IS SMALL (BVH depth <= 16, the Cornell box): the deep bounces hold a few dozen rays per
    // segment and every visit costs a chunk's latency, so fewer segments and the static stride win (800^2 x 32 spp: 16.3 -> 14.4 ms per
    // frame; x 64: 29.1 -> 27.8; x 128 and x 256 prefer the tickets).  The 10^6-triangle scene's rays are too uneven for a static stride
    // at any size (1024^2 x 32 spp: 115 -> 128 ms).
    long mid_paths = 48L << 20;
    if (const char* e = hk::knob("HK_MID_PASS_PATHS_M")) mid_paths = std::atol(e) >= 0 ? std::atol(e) << 20 : mid_paths;
    const bool mid = !media && !open_scene && small_scene && (long)capacity <= mid_paths;
    I->mid_pass = mid;
    const long cap_per_cu = media ? 96 : ((open_scene || mid) ? 48 : 192);
    long W_want = ((long)(capacity + 63) / 64) / 16;
    if (W_want < 4L * n_cu) W_want = 4L * n_cu;
    if (W_want > cap_per_cu * n_cu) W_want = cap_per_cu * n_cu;
    {   // read per call (tests toggle it); an unset knob is the built-in policy again — not the last forced count
        const char* e = hk::knob("HK_WAVES_PER_CU");
        I->ctx->waves_per_cu = e && std::atoi(e) > 0 ? std::atoi(e) : 0;
    }
    if (I->ctx->waves_per_cu > 0) W_want = (long)I->ctx->waves_per_cu * n_cu;
    W_want = (W_want + 3) / 4 * 4;
    I->st.dynamic_segments = (media || (!open_scene && !mid)) ? 1 : 0;
    if (const char* e = hk::knob("HK_DYNAMIC_SEGMENTS")) I->st.dynamic_segments = std::atoi(e) ? 1 : 0;
    I->st.compact = media ? (grey_compact ? 1 : 0) : 1;
    {   // slim shadow records (HK_SHADOW_FINAL=0: the 60-byte records of rounds 3-5; A/B switch, films bit-identical)
        const char* e = hk::knob("HK_SHADOW_FINAL");
        I->st.sh_final = (!media && opaque_surfaces && !(e && std::atoi(e) == 0)) ? 1 : 0;   // (k_shadow's scenes: no medium, no alpha-tested surface)
        // lean generation records of the same scenes (HK_LEAN_RECORDS=0: 8-byte meta words, depth-0 origins stored; A/B switch, films bit-identical)
        const char* l = hk::knob("HK_LEAN_RECORDS");
        I->st.meta32 = (!media && opaque_surfaces && !(l && std::atoi(l) == 0)) ? 1 : 0;
        I->st.const_origin = 0;   // (per call: render_tile_now knows the camera)
    }
    I->st.ticket_share = 1;
    if (const char* e = hk::knob("HK_TICKET_SHARE")) I->st.ticket_share = std::atoi(e) ? 1 : 0;
    {   // a SMALL pass (at most 16 chunks for each of 4 waves per CU: a one-sample call at 800^2) hands segment g to wave g of every launch
        // (static stride where a grid is smaller): the 17 k_segment_lists launches of a call and the ticket round trips go, and the segment
        // count no longer has to stay below every kernel's residency, so it can be as fine as the latency of one wave's chunk wants it —
        // HK_SMALL_PASS_WAVES per CU.  HK_SMALL_PASS=0: lists and tickets as ever.
        const char* e = hk::knob("HK_SMALL_PASS");
        const bool small = W_want <= 4L * n_cu && !(e && std::atoi(e) == 0) && hk::knob("HK_DYNAMIC_SEGMENTS") == nullptr;   // (an explicit HK_DYNAMIC_SEGMENTS keeps lists / tickets: the tests' small films)
        I->st.small_pass = small ? 1 : 0;
        if (mid && I->st.dynamic_segments == 0 && !(e && std::atoi(e) == 0) && !(hk::knob("HK_MID_LISTS") && std::atoi(hk::knob("HK_MID_LISTS"))))
            I->st.small_pass = 1;   // a mid-size pass of a closed scene: its 48 segments per CU all hold paths down to the last bounce — the work lists (17 launches) list everything
        if (small) {
            I->st.dynamic_segments = 0;
            const char* w = hk::knob("HK_SMALL_PASS_WAVES");
            // (800^2, one sample per call.  Eight calls in flight on eight lanes (HK_PIPELINE): Cornell 0.99 / 0.89 / 1.03 / 1.30 ms per call at
            // 4 / 8 / 16 / 32 segments per CU, cloud 12.6 / 13.8 / 17.9 / 19.6.  ONE call at a time — the default since small calls are batched,
            // and what a caller that looks at every frame gets: Cornell 2.04 / 1.53 / 1.41 / 1.34 / 1.59 / 1.63 ms at 4 / 8 / 12 / 16 / 20 / 32,
            // cloud 29.8 / 26.0 / 25.4 at 4 / 8 / 16, 10^6 triangles 11.0 / 10.9 / 12.1 at 8 / 16 / 32, sky 2.32 / 2.39 / 2.49: one segment per
            // resident wave — 16 waves per CU is what k_trace_lean's 16-wave block and k_shade's 128 registers hold, and at 8 the 1024-thread
            // trace blocks covered half of the CUs.  An OPEN surface scene keeps 8: its deep bounces are empty and its glass / conductor
            // shade kernels hold fewer than 16 waves per CU.)
            // k_small_pass (`fusable`: the whole pass in one launch) likes 8: 1.05 ms against 1.14 at 16 — fuller chunks, no spills at two waves per SIMD.
            long per_cu = w && std::atoi(w) > 0 ? std::atoi(w) : (users ? (media ? 4 : 8) : (((open_scene && !media) || fusable) ? 8 : 16));
            const long chunks = ((long)capacity + 63) / 64;
            while (per_cu > 4 && per_cu * n_cu > chunks) per_cu /= 2;   // (no segment without a chunk)
            if (I->ctx->waves_per_cu <= 0) W_want = (per_cu * n_cu + 3) / 4 * 4;
        }
    }
    const char* split_env = hk::knob("HK_WALK_SPLIT");
    const bool want_split = media && split_env && std::atoi(split_env);
    // the retained state must be of the same flavour: a state allocated for a scene with media has no sel_light (k_light_select would
    // silently not run in a later scene without media), one allocated without HK_WALK_SPLIT has no hand-over queues
    if (I->st_capacity >= capacity && I->st_depth >= I->p.max_depth && I->st.n_waves == (int)W_want && I->st_media == (media ? 1 : 0) && (!want_split || I->st.wq_a != nullptr)) return HK_OK;
    if (users) HIP_TRY(hipStreamSynchronize(users));   // (a lane's set: its last call may still be running)
    if (!I->bufs.empty()) quiesce(I->ctx);             // the arrays about to be let go (a cached slab is handed on without a device-wide wait)
    for (auto* b : I->bufs) delete b;
    I->bufs.clear();
    DPathState& s = I->st;
    size_t P = (size_t)capacity;
    const int W = (int)W_want;
    const int chunks = (capacity + 63) / 64;
    s.capacity = capacity;
    s.n_waves = W;
    s.wave_cap = ((chunks + W - 1) / W) * 64;
    const size_t Q = (size_t)W * s.wave_cap;
    auto layout = [&]() -> int {
        for (int gidx = 0; gidx < 2; ++gidx) {   // two generations of path records in queue order (DPathGen)
            DPathGen& g = s.gen[gidx];
            HIP_TRY(alloc_arr(I, g.ray_o, Q));
            HIP_TRY(alloc_arr(I, g.ray_d, Q));
            HIP_TRY(alloc_arr(I, g.beta, Q));
            HIP_TRY(alloc_arr(I, g.r_u, Q));
            HIP_TRY(alloc_arr(I, g.r_l, Q));
    #if HK_LAMBDA_BY_SLOT
            g.lambda = nullptr;
    #else
            HIP_TRY(alloc_arr(I, g.lambda, Q));
    #endif
            HIP_TRY(alloc_arr(I, g.meta, Q));
        }
        HIP_TRY(alloc_arr(I, s.hit, Q));
        HIP_TRY(alloc_arr(I, s.mat_id, Q));
        s.sel_light = nullptr;
        if (!media) HIP_TRY(alloc_arr(I, s.sel_light, Q));   // k_light_select's results (8 B per entry)
        HIP_TRY(alloc_arr(I, s.lambda_s, P));
        s.pdf = nullptr;   // recomputed from lambda_s by k_film
        HIP_TRY(alloc_arr(I, s.L, P));
        HIP_TRY(alloc_arr(I, s.filter_w, P));
        HIP_TRY(alloc_arr(I, s.sh_o, Q));
        HIP_TRY(alloc_arr(I, s.sh_d, Q));
        HIP_TRY(alloc_arr(I, s.sh_Ld, Q));
        HIP_TRY(alloc_arr(I, s.sh_ru, Q));
        HIP_TRY(alloc_arr(I, s.sh_rl, Q));
        HIP_TRY(alloc_arr(I, s.sh_slot, Q));
        s.sh_T = nullptr, s.sh_aux = nullptr, s.sh_it = nullptr, s.wq_a = nullptr, s.wq_b = nullptr, s.wq_ctl = nullptr;
        if (want_split) {   // the split shadow walk of grey media (92 B per record; off by default)
            HIP_TRY(alloc_arr(I, s.sh_T, Q));
            HIP_TRY(alloc_arr(I, s.sh_aux, Q));
            HIP_TRY(alloc_arr(I, s.sh_it, 4 * Q));
            // worst case of gq_out_push: a chunk of 256 entries is closed (padded) as soon as the next push of up to 64 does not fit, so 193
            // pushes can use up 256 entries, plus one open chunk per resident wave (at most 32 waves per CU)
            const size_t wq_n = (Q / 193 + 1) * 256 + (size_t)256 * 32 * (size_t)n_cu;
            HIP_TRY(alloc_arr(I, s.wq_a, wq_n));
            HIP_TRY(alloc_arr(I, s.wq_b, wq_n));
            HIP_TRY(alloc_arr(I, s.wq_ctl, (size_t)(I->p.max_depth + 2) * 11 * 4));
        }
        HIP_TRY(alloc_arr(I, s.escaped_q, Q));
        HIP_TRY(alloc_arr(I, s.medium_q, Q));
        HIP_TRY(alloc_arr(I, s.scatter_q, Q));
        s.ticket_rows = I->p.max_depth + 2;
        HIP_TRY(alloc_arr(I, s.tickets, (size_t)s.ticket_rows * HK_TICKET_COLS * HK_TICKET_WAYS * HK_TICKET_STRIDE));
        HIP_TRY(alloc_arr(I, s.initial_medium, 1));
        if (I->slab_mode != 1) HIP_TRY(hipMemset(s.initial_medium, 0xff, sizeof(int)));
        HIP_TRY(alloc_arr(I, s.mat_q, Q * HK_MAX_KINDS));
        size_t nc = (size_t)(I->p.max_depth + 2) * Q_COUNT * W;
        HIP_TRY(alloc_arr(I, s.counters, nc));
        if (I->slab_mode != 1) HIP_TRY(hipMemset(s.counters, 0, nc * sizeof(int)));
        HIP_TRY(alloc_arr(I, s.seg_list, nc));
        HIP_TRY(alloc_arr(I, s.seg_list_n, (size_t)(I->p.max_depth + 2) * Q_COUNT));
        return HK_OK;
    };
    const char* slab_env = hk::knob("HK_STATE_SLAB");   // (0: one allocation per array, nothing cached)
    if (!(slab_env && std::atoi(slab_env) == 0)) {
        I->slab_mode = 1, I->slab_off = 0;
        if (int e = layout()) {
            I->slab_mode = 0;
            return e;
        }
        DevBuf* slab = new DevBuf();
        I->bufs.push_back(slab);
        const hipError_t he = slab->alloc_slab(I->ctx->device, I->slab_off);
        if (he != hipSuccess) {
            I->slab_mode = 0;
            return fail(HK_ERR_DEVICE, "path-state slab allocation failed");
        }
        I->slab_base = slab->p;
        I->slab_mode = 2, I->slab_off = 0;
        const int e = layout();
        I->slab_mode = 0;
        if (e) return e;
    } else if (int e = layout())
        return e;
    I->st_capacity = capacity;
    I->st_depth = I->p.max_depth;
    I->st_media = media ? 1 : 0;
    if (hk::knob("HK_DEBUG_ALLOC"))
        std::fprintf(stderr, "HK_DEBUG_ALLOC capacity %d Q %zu: ray_o %p ray_d %p beta %p hit %p sh_o %p L %p medium_q %p mat_q %p counters %p\n", capacity, Q, (void*)s.gen[0].ray_o,
                     (void*)s.gen[0].ray_d, (void*)s.gen[0].beta, (void*)s.hit, (void*)s.sh_o, (void*)s.L, (void*)s.medium_q, (void*)s.mat_q, (void*)s.counters);
    return HK_OK;
}
int ceil_log2(long v) {
    int l = 0;
    while ((1L << l) < v) ++l;
    return l;
}
DSobol make_sobol(const hk_integrator_params& p, int w, int h) {  // compute_zsobol_params, sobol.jl:317-323; volpath.jl:475
    DSobol s;
    int spp = p.samples_per_pixel > 4096 ? p.samples_per_pixel : 4096;
    s.log2_spp = ceil_log2(spp < 1 ? 1 : spp);
    int res_log2 = ceil_log2(w > h ? w : h);
    s.n_base4_digits = res_log2 + (s.log2_spp + 1) / 2;
    s.seed = p.sampler_seed;
    s.width = w;
    s.hi_table = nullptr;
    s.hi_rows = s.hi_stride = 0;
    s.lo_table = nullptr;
    s.lo_rows = s.lo_count = s.lo_offset = 0;
    return s;
}
DCamera make_camera(const hk_camera& c) {
    DCamera d;
    std::memcpy(d.r2c, c.raster_to_camera, 64);
    std::memcpy(d.c2w, c.camera_to_world, 64);
    d.lens_radius = c.lens_radius;
    d.focal_distance = c.focal_distance;
    d.shutter_open = c.shutter_open;
    d.shutter_close = c.shutter_close;
    return d;
}
hipEvent_t get_event(hk_ctx* c) {
    if (!c->event_pool.empty()) {
        hipEvent_t e = c->event_pool.back();
        c->event_pool.pop_back();
        return e;
    }
    hipEvent_t e;
    (void)hipEventCreate(&e);
    return e;
}
}  // namespace

extern "C" int32_t hk_render(hk_ctx* c, hk_scene* sc, hk_integrator* I, hk_film* film, const hk_camera* cam, int32_t first_sample_idx, int32_t n_samples,
                             int32_t sample_stride) {
    if (!film) return fail(HK_ERR_INVALID, "null argument");
    return hk_render_tile(c, sc, I, film, cam, first_sample_idx, n_samples, sample_stride, 0, 0, film->width, film->height);
}

static int render_tile_now(hk_ctx* c, hk_scene* sc, hk_integrator* I, hk_film* film, const hk_camera* cam, int32_t first_sample_idx, int32_t n_samples,
                           int32_t sample_stride, int32_t x0, int32_t y0, int32_t x1, int32_t y1);
static int flush_pending(hk_ctx* c) {
    if (!c || !c->pending.active) return HK_OK;
    hk_ctx::Pending p = c->pending;
    c->pending.active = false;
    return render_tile_now(c, p.sc, p.I, p.film, &p.cam, p.first, p.n, p.stride, p.x0, p.y0, p.x1, p.y1);
}
extern "C" int32_t hk_render_tile(hk_ctx* c, hk_scene* sc, hk_integrator* I, hk_film* film, const hk_camera* cam, int32_t first_sample_idx, int32_t n_samples,
                                  int32_t sample_stride, int32_t x0, int32_t y0, int32_t x1, int32_t y1) {
    if (!c || !sc || !I || !film || !cam) return fail(HK_ERR_INVALID, "null argument");
    if (x0 < 0 || y0 < 0 || x1 > film->width || y1 > film->height || x0 > x1 || y0 > y1) return fail(HK_ERR_INVALID, "pixel range outside the film");
    if (x0 == x1 || y0 == y1) return HK_OK;
    if (n_samples < 0 || sample_stride < 1 || first_sample_idx < 1) return fail(HK_ERR_INVALID, "bad sample range");
    if (!c->have_tables) return fail(HK_ERR_INVALID, "hk_ctx_set_tables must be called first");
    if (sc->ctx != c || I->ctx != c || film->ctx != c) return fail(HK_ERR_INVALID, "scene / integrator / film belong to another context");
    if (film->f64 != (I->p.accumulate_f64 != 0)) return fail(HK_ERR_INVALID, "film / integrator accumulation type mismatch");
    if (n_samples == 0) return HK_OK;
    KnobScope knobs(&c->knobs);
    // A small call (a one-sample `render!`: < 1 path per resident lane, 26 launches of 26 - 100 us whatever they hold) is worth a seventh
    // of its time as part of a larger pass, and the film does not depend on the pass size (k_film adds in sample order).  So a small call
    // is only NOTED; calls that continue it grow the note; the pass is rendered when the note holds HK_BATCH_PATHS_M (64) million paths,
    // when a call comes that does not continue it, or when anything looks at the result (flush_pending).  Rendering is asynchronous
    // either way: a device error of a deferred pass is reported by the call that flushes it.  HK_BATCH_PATHS_M=0: every call at once.
    {
        long batch_paths = 64L << 20, small_paths = 8L << 20;   // (Cornell 800^2, 64 one-sample calls: 0.49 / 0.46 / 0.45 ms per call at 32 / 48 / 64 M, cloud 3.8 / 3.7 / 2.8)
        if (const char* e = hk::knob("HK_BATCH_PATHS_M")) batch_paths = std::atol(e) >= 0 ? std::atol(e) << 20 : batch_paths;
        if (const char* e = hk::knob("HK_PIPELINE_MAX_PATHS_M")) small_paths = std::atol(e) > 0 ? std::atol(e) << 20 : small_paths;
        const long px = (long)((x1 - x0 + 7) / 8) * ((y1 - y0 + 7) / 8) * 64;
        hk_ctx::Pending& p = c->pending;
        // A caller that owns the stream or the accumulators may order its own work behind this call with stream / event calls the library
        // never sees (torch.cuda.synchronize(), a reduce of the accumulators): its calls are enqueued at once.  HK_DEFER_EXTERNAL=1: batched all the same.
        const char* de = hk::knob("HK_DEFER_EXTERNAL");
        const bool visible_order = (c->own_stream_order && !film->external && !film->exposed) || (de && std::atoi(de));
        const bool small = batch_paths > 0 && visible_order && !c->time_kernels && (long)n_samples * px <= small_paths && I->p.samples_per_pass <= 0;
        if (p.active && small && p.sc == sc && p.I == I && p.film == film && std::memcmp(&p.cam, cam, sizeof(hk_camera)) == 0 && p.stride == sample_stride && p.x0 == x0 &&
            p.y0 == y0 && p.x1 == x1 && p.y1 == y1 && (long)p.first + (long)p.n * p.stride == first_sample_idx && (long)(p.n + n_samples) * px <= batch_paths) {
            p.n += n_samples;
            p.calls += 1;
            return HK_OK;
        }
        if (int e = flush_pending(c)) return e;
        if (small) {
            p.active = true;
            p.sc = sc, p.I = I, p.film = film, p.cam = *cam;
            p.first = first_sample_idx, p.n = n_samples, p.stride = sample_stride, p.x0 = x0, p.y0 = y0, p.x1 = x1, p.y1 = y1, p.calls = 1;
            return HK_OK;
        }
    }
    return render_tile_now(c, sc, I, film, cam, first_sample_idx, n_samples, sample_stride, x0, y0, x1, y1);
}
static int render_tile_now(hk_ctx* c, hk_scene* sc, hk_integrator* I, hk_film* film, const hk_camera* cam, int32_t first_sample_idx, int32_t n_samples,
                           int32_t sample_stride, int32_t x0, int32_t y0, int32_t x1, int32_t y1) {
    HIP_TRY(hipSetDevice(c->device));
    KnobScope knobs(&c->knobs);   // (flush_pending reaches this from every entry point)
    const int W = film->width, H = film->height;
    DFrame fr{};
    fr.width = W;
    fr.height = H;
    fr.x0 = x0, fr.y0 = y0, fr.x1 = x1, fr.y1 = y1;
    fr.tiles_x = (x1 - x0 + 7) / 8;
    fr.tiles_y = (y1 - y0 + 7) / 8;
    fr.n_pixels_padded = fr.tiles_x * fr.tiles_y * 64;
    int S = I->p.samples_per_pass;
    if (S <= 0) {
        // auto: up to ~192 M paths in flight (256 spp of an 800x800 frame).  The chip holds 256 CUs x 16..32 waves x 64 lanes, and
        // the deeper bounces of a pass only keep it busy when the pass starts with hundreds of paths per lane (Cornell 800^2: 5.9 G
        // rays/s at 32 spp per pass, 6.3 at 64, 6.45 at 128, 6.56 at 256); path state is ~440 B per path, i.e. up to ~80 GB of the
        // 288 GB of HBM, halved until it fits in half of the free memory.  HK_MAX_PATHS_M overrides the cap (millions of paths).
        long max_paths = 192L << 20;
        if (const char* e = hk::knob("HK_MAX_PATHS_M")) max_paths = std::atol(e) > 0 ? std::atol(e) << 20 : max_paths;
        S = (int)(max_paths / fr.n_pixels_padded);
        if (S < 1) S = 1;
        if (S > 256) S = 256;
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
            free_b += g_slabs.total(c->device);   // (cached path-state slabs are given back when an allocation needs them)
            while (S > 1 && (size_t)S * fr.n_pixels_padded * 440 > free_b / 2) S /= 2;
        }
    }
    if (S > n_samples) S = n_samples;
    if ((long)S * fr.n_pixels_padded > 0x3fffffffL) return fail(HK_ERR_INVALID, "pass too large");
    // a small one-pass call goes to the next LANE (hk_ctx::Lane): it runs beside the small calls before and after it
    int lane_idx = -1;
    {
        // measured (Cornell 800^2, one sample per call, ms per call): 1 lane 2.47, 2 lanes 1.69, 4 lanes 1.66, 8 lanes 1.36, 16 lanes 1.21;
        // the lanes' path-state sets together stay below 8 GB (440 B per path)
        int n_lanes = 1;   // (off since small calls are batched — HK_PIPELINE = 2 .. 16 turns the lanes on for calls that cannot be: a read-back between them)
        long max_paths = 8L << 20;
        if (const char* e = hk::knob("HK_PIPELINE")) n_lanes = std::atoi(e) >= 1 && std::atoi(e) <= 16 ? std::atoi(e) : n_lanes;
        if (const char* e = hk::knob("HK_PIPELINE_MAX_PATHS_M")) max_paths = std::atol(e) > 0 ? std::atol(e) << 20 : max_paths;
        {
            const long by_memory = (8L << 30) / (440L * (long)S * fr.n_pixels_padded);
            if (n_lanes > by_memory) n_lanes = by_memory < 1 ? 1 : (int)by_memory;
        }
        // (the lanes are streams of their own — see the note on the second hardware queue at `overlap` below — so one small call, or a
        // handful, does not start them: only a run of HK_PIPELINE_AFTER = 4 small calls in a row does, i.e. a progressive viewer)
        const bool small_call = n_samples <= S && (long)S * fr.n_pixels_padded <= max_paths;
        int after = 4;
        if (const char* e = hk::knob("HK_PIPELINE_AFTER")) after = std::atoi(e) >= 0 ? std::atoi(e) : after;
        c->small_streak = small_call ? c->small_streak + 1 : 0;
        const bool lanes_started = c->ev_main != nullptr;
        if (n_lanes > 1 && !c->time_kernels && small_call && (lanes_started || c->small_streak > after)) {
            if ((int)I->lane_sets.size() < (int)hk_ctx::MAX_LANES) I->lane_sets.resize(hk_ctx::MAX_LANES);
            lane_idx = c->next_lane % n_lanes;
            c->next_lane = (lane_idx + 1) % n_lanes;
            hk_ctx::Lane& L = c->lanes[lane_idx];
            if (!L.stream) HIP_TRY(hipStreamCreateWithFlags(&L.stream, hipStreamNonBlocking));
            if (!L.done) {
                HIP_TRY(hipEventCreateWithFlags(&L.done, hipEventDisableTiming));
                std::vector<DStats> zero((size_t)c->stat_rows);
                std::memset(zero.data(), 0, zero.size() * sizeof(DStats));
                HIP_TRY(L.stats.upload(zero.data(), zero.size() * sizeof(DStats)));
            }
            if (!c->ev_main) {
                HIP_TRY(hipEventCreateWithFlags(&c->ev_main, hipEventDisableTiming));
                HIP_TRY(hipEventCreateWithFlags(&c->ev_film, hipEventDisableTiming));
            }
        } else {
            if (int e = join_lanes(c)) return e;
        }
    }
    const bool piped = lane_idx >= 0;
    // the lane's path-state set stands in for the integrator's own for the duration of this call
    struct SetGuard {
        hk_integrator* I;
        hk_integrator::StateSet* ls;
        ~SetGuard() {
            if (ls) I->swap_set(*ls);
        }
    } set_guard{I, piped ? &I->lane_sets[lane_idx] : nullptr};
    if (piped) I->swap_set(I->lane_sets[lane_idx]);
    int st = ensure_state(I, S * fr.n_pixels_padded, sc->d.n_media > 0, sc->d.has_escape_lights != 0, sc->d.bvh_depth <= 16, sc->d.n_media > 0 && hk::grey_compact_ok(sc->d), piped ? c->lanes[lane_idx].stream : nullptr,
                          // (the predicate of launch_small_pass, as far as it is known here: a call of >= 16 samples draws from the sample-bit table
                          // and keeps the launches — those want 16 segments per CU, not k_small_pass's 8)
                          !piped && !c->time_kernels && !c->count_nodes && n_samples < 16 && hk::small_pass_fusable(sc->d, sc->kinds_mask),
                          sc->d.all_opaque != 0 && sc->d.n_media == 0);
    if (st != HK_OK) return st;
    I->st.const_origin = (I->st.meta32 && !(cam->lens_radius > 0)) ? 1 : 0;   // a pinhole camera: generate_ray gives every path the same origin (hk_device.h), stored once per segment
    fr.sample_stride = sample_stride;
    fr.max_depth = I->p.max_depth;
    fr.regularize = I->p.regularize;
    fr.max_component_value = I->p.max_component_value;
    fr.count_nodes = c->count_nodes;
    fr.implicit_ones = sc->d.n_media == 0 ? 1 : 0;
    {   // loop shapes of the tracking state machines (measured defaults; read per call)
        auto knob = [](const char* name, int dflt) {
            const char* e = hk::knob(name);
            const int v = e ? std::atoi(e) : dflt;
            return v >= 1 && v <= 255 ? v : dflt;
        };
        {   // collision rounds of k_track_flat wait for HK_TRACK_MIN_PENDING lanes, for at most HK_TRACK_EXTRA_ADVANCE further cheap steps (0 = never wait)
            const char* e = hk::knob("HK_TRACK_MIN_PENDING");
            const char* x = hk::knob("HK_TRACK_EXTRA_ADVANCE");
            const int mp = e ? std::atoi(e) : 0, xa = x ? std::atoi(x) : 4;
            fr.track_gate = (mp >= 0 && mp <= 64 ? mp : 0) | ((xa >= 0 && xa <= 255 ? xa : 4) << 8);
        }
        fr.delta_advance = knob("HK_DELTA_ADVANCE", 3);
        fr.refill_idle = knob("HK_TRACK_REFILL_IDLE", 24);
        fr.walk_tune = knob("HK_SHADOW_TRACK_BATCH", 8) | (knob("HK_TRACK_ADVANCE", 2) << 8) | (knob("HK_SHADOW_FEED_ROUNDS", 3) << 16) | ((knob("HK_WALK_REFILL_IDLE", 16) > 63 ? 63 : knob("HK_WALK_REFILL_IDLE", 16)) << 24);
    }
    DSobol sob = make_sobol(I->p, W, H);
    {   // pixel-digit table of the sampler: depends on film size, spp exponent and max_depth only
        const int rows = 4 + 5 * (I->p.max_depth + 1);
        const bool table_same = !(I->sobol_rows != rows || I->sobol_stride != fr.n_pixels_padded || I->sobol_log2 != sob.log2_spp || I->sobol_digits != sob.n_base4_digits ||
                                  I->sobol_x0 != x0 || I->sobol_y0 != y0 || I->sobol_tiles_x != fr.tiles_x);
        if (!table_same) {
            if (c->lanes_dirty) {   // lanes may still read the table that is about to be replaced
                if (int e = join_lanes(c)) return e;
                HIP_TRY(hipStreamSynchronize(c->stream));
            }
            HIP_TRY(I->sobol_table.alloc((size_t)rows * fr.n_pixels_padded * sizeof(uint2)));
            hk::launch_sobol_table(c->stream, sob, fr, I->sobol_table.as<uint2>(), rows);
            HIP_TRY(hipGetLastError());
            I->sobol_rows = rows;
            I->sobol_stride = fr.n_pixels_padded;
            I->sobol_log2 = sob.log2_spp;
            I->sobol_digits = sob.n_base4_digits;
            I->sobol_x0 = x0, I->sobol_y0 = y0, I->sobol_tiles_x = fr.tiles_x;
        }
        sob.hi_table = I->sobol_table.as<uint2>();
        sob.hi_rows = rows;
        sob.hi_stride = fr.n_pixels_padded;
        // sample-bit table: worth building when the call draws many samples per pixel (a frame), not for one-sample progressive calls.
        // Two bytes per (row, pixel, sample): rows are cut to HK_SOBOL_LO_GB (default 32 GB, and a quarter of the free memory) —
        // the deep rows serve few paths and fall back to hashing the sample digits.
        const long last_idx = (long)first_sample_idx + (long)(n_samples - 1) * sample_stride;
        if (n_samples >= 16 && sob.log2_spp >= 2 && sob.log2_spp <= 16 && (last_idx >> sob.log2_spp) == 0) {
            const int base = sample_stride == 1 ? (first_sample_idx & ~15) : first_sample_idx;
            const int count = (int)((((last_idx - base) / sample_stride + 1) + 15) & ~15L);
            bool same = table_same && I->lo_base == base && I->lo_sample_stride == sample_stride && I->lo_count == count && I->lo_rows > 0;
            if (!same) {
                if (c->lanes_dirty) {   // lanes may still draw from the table that is rebuilt in the same buffer (as for the hi table above)
                    if (int e = join_lanes(c)) return e;
                    HIP_TRY(hipStreamSynchronize(c->stream));
                }
                double gb = 32.0;
                if (const char* e = hk::knob("HK_SOBOL_LO_GB")) gb = std::atof(e);
                size_t budget = (size_t)(gb * 1e9), free_b = 0, total_b = 0;
                // the buffer of the previous table is reused when it is large enough (another sample range of the same film)
                if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && (free_b + I->sobol_lo.bytes) / 4 < budget) budget = (free_b + I->sobol_lo.bytes) / 4;
                const size_t per_row = (size_t)fr.n_pixels_padded * count * sizeof(uint16_t);
                int lo_rows = (int)(budget / per_row < (size_t)rows ? budget / per_row : (size_t)rows);
                I->lo_rows = 0;
                if (lo_rows > 0 && (size_t)fr.n_pixels_padded * count < 0xffffffffull) {
                    // the table is an optimisation: when its memory cannot be had (another context on the device took it since
                    // hipMemGetInfo), the call renders with hashed digits instead of failing
                    bool have_buf = true;
                    if (I->sobol_lo.bytes < per_row * lo_rows || I->sobol_lo.bytes > 2 * per_row * lo_rows) {
                        while (lo_rows > 0 && I->sobol_lo.alloc(per_row * lo_rows) != hipSuccess) {
                            (void)hipGetLastError();
                            lo_rows /= 2;
                        }
                        have_buf = lo_rows > 0;
                    }
                    if (have_buf) {
                        hk::launch_sobol_lo_table(c->stream, sob, fr, I->sobol_lo.as<uint16_t>(), lo_rows, base, sample_stride, count);
                        HIP_TRY(hipGetLastError());
                        I->lo_rows = lo_rows;
                        I->lo_base = base;
                        I->lo_sample_stride = sample_stride;
                        I->lo_count = count;
                    }
                }
            }
            if (I->lo_rows > 0) {
                sob.lo_table = I->sobol_lo.as<uint16_t>();
                sob.lo_rows = I->lo_rows;
                sob.lo_count = I->lo_count;
            }
        }
    }
    DCamera dc = make_camera(*cam);
    DStats* dstats = c->stats.as<DStats>();
    const int trace_blocks = c->n_cu, shade_blocks = c->n_cu, light_blocks = c->n_cu;  // launchers size the grid from residency
    hipStream_t s = c->stream;
    if (piped) {   // behind everything the context's stream holds so far (scene upload, film clear, sampler tables), beside the other lanes
        HIP_TRY(hipEventRecord(c->ev_main, c->stream));
        s = c->lanes[lane_idx].stream;
        HIP_TRY(hipStreamWaitEvent(s, c->ev_main, 0));
        dstats = c->lanes[lane_idx].stats.as<DStats>();
    }
    if (!c->have_span) {
        HIP_TRY(hipEventRecord(c->ev_begin, s));
        c->have_span = true;
    }
    // K14: camera medium, decided on the device (no readback)
    if (sc->d.n_media > 0) {
        const float* m = cam->camera_to_world;
        float w = m[15];
        float cx = m[3], cy = m[7], cz = m[11];
        if (w != 1.0f) {
            float inv = 1.0f / w;
            cx *= inv;
            cy *= inv;
            cz *= inv;
        }
        hk::launch_detect_camera_medium(s, I->st, sc->d, cx, cy, cz, dstats);
    } else
        HIP_TRY(hipMemsetAsync(I->st.initial_medium, 0xff, sizeof(int), s));
    int done = 0;
    while (done < n_samples) {
        int k = n_samples - done < S ? n_samples - done : S;
        fr.samples_in_pass = k;
        {   // reciprocal of the pass's sample count: M = ceil(2^(30+L) / k), L = ceil(log2 k); exact for slots below 2^30 (checked above)
            int L = 0;
            while ((1 << L) < k) ++L;
            fr.s_shr = 30 + L;
            fr.s_mul = (uint32_t)((((uint64_t)1 << fr.s_shr) + (uint64_t)k - 1) / (uint64_t)k);
        }
        fr.first_sample = first_sample_idx + done * sample_stride;
        if (sob.lo_table) sob.lo_offset = (fr.first_sample - I->lo_base) / sample_stride;
        auto timed = [&](int cls, auto&& fn) -> int {
            hipEvent_t e0 = nullptr, e1 = nullptr;
            if (c->time_kernels) {
                e0 = get_event(c);
                e1 = get_event(c);
                if (hipEventRecord(e0, s) != hipSuccess) return HK_ERR_DEVICE;
            }
            fn();
            if (c->time_kernels) {
                if (hipEventRecord(e1, s) != hipSuccess) return HK_ERR_DEVICE;
                (cls == 0 ? c->trace_events : c->class_events[cls]).emplace_back(e0, e1);
            }
            return HK_OK;
        };
        // a small pass of a closed all-matte scene: camera rays and every bounce in ONE launch (k_small_pass), then the film.  It takes no
        // tickets, and every queue size it reads was written before by the same wave: the three clears below are not for it.
        const bool fused = !c->time_kernels && !piped && hk::launch_small_pass(s, c->n_cu, I->st, sc->d, c->tables, fr, I->filter, dc, sob, I->p.max_depth, sc->kinds_mask, dstats, true, nullptr, 0);
        if (!fused) {
        HIP_TRY(hipMemsetAsync(I->st.tickets, 0, (size_t)I->st.ticket_rows * HK_TICKET_COLS * HK_TICKET_WAYS * HK_TICKET_STRIDE * sizeof(int), s));
        // queue sizes start every pass at zero: a depth at which no shade / scatter kernel runs (a triangle-free scene lit by an
        // environment map, say) must not see the ray / shadow counts an earlier render left behind
        HIP_TRY(hipMemsetAsync(I->st.counters, 0, (size_t)(I->st_depth + 2) * Q_COUNT * I->st.n_waves * sizeof(int), s));
        if (I->st.wq_ctl) HIP_TRY(hipMemsetAsync(I->st.wq_ctl, 0, (size_t)(I->st_depth + 2) * 11 * 4 * sizeof(int), s));
        }
        bool film_inline = false;
        if (fused) {
            film_inline = fr.samples_in_pass == 1;   // a one-sample pass adds its paths to the film itself
            (void)hk::launch_small_pass(s, c->n_cu, I->st, sc->d, c->tables, fr, I->filter, dc, sob, I->p.max_depth, sc->kinds_mask, dstats, false, film->accum, film_inline ? (film->f64 ? 2 : 1) : 0);
            c->fused_passes++;
        }
        if (!fused) {
        if (timed(3, [&] { hk::launch_camera(s, c->n_cu, I->st, fr, c->tables, I->filter, dc, sob, -1); }) != HK_OK) return fail(HK_ERR_DEVICE, "event record failed");
        // work lists: after every producer, the non-empty segments of the queues it filled (consumers never visit an empty segment)
        auto lists = [&](std::initializer_list<std::pair<int, int>> dq, bool kinds_of_depth = false, int kd = 0) -> int {
            int dd[HK_MAX_KINDS + 6], qq[HK_MAX_KINDS + 6], n = 0;
            for (auto& e : dq) dd[n] = e.first, qq[n] = e.second, ++n;
            if (kinds_of_depth)
                for (int kind = 0; kind < HK_MAX_KINDS; ++kind)
                    if (sc->kinds_mask & (1u << kind)) dd[n] = kd, qq[n] = Q_MAT0 + kind, ++n;
            if (I->st.small_pass) return HK_OK;
            return timed(3, [&] { hk::launch_segment_lists(s, I->st, n, dd, qq); });
        };
        if (lists({{0, Q_RAY}}) != HK_OK) return fail(HK_ERR_DEVICE, "event record failed");
        // Two streams: k_shadow of bounce d touches only its shadow records and L; k_trace of bounce d + 1 touches neither.  The
        // shadow launch goes to a second stream behind the shade kernels, the next traversal starts beside it, and the first stream
        // waits for the shadows before anything that adds to L or rewrites shadow records (media tracking, escaped, shade).  The two
        // fill each other's tails and stalls.  With per-kernel timing on (hk_stats_enable_counters bit 1) everything stays on one
        // stream so that the class times add up.
        // (surfaces only: beside the long shadow walks of a media scene the next traversal only competes — cloud -4.5 %; Cornell +-0, sky +1 %, many-light +2.4 %)
        if (const char* e = hk::knob("HK_OVERLAP")) c->overlap = std::atoi(e) ? 1 : 0;
        else c->overlap = -1;
        // A SECOND HARDWARE QUEUE IS NOT FREE ON THIS CHIP: from the moment a process has used a second stream of its own, every kernel of
        // every stream takes 50 - 120 us longer (the completion of one and the start of the next, with or without events between the
        // streams, whatever their flags and priorities; GPU_MAX_HW_QUEUES=1 makes it go away, and so does never creating the stream) —
        // the "two speeds" of DESIGN.md §5: 8 % of a cloud frame (330 launches), 1 - 4 % of a Cornell frame.  The shadow launch on a
        // second stream repays that only where launches are long: the 10^6-triangle scene (+2.9 %); Cornell and sky +-0.1 % with
        // outliers of +4 %.  So: automatic for deep BVHs only, and never in a mid-size pass (800^2 x 32 spp: -4 %).
        const bool want_overlap = c->overlap < 0 ? sc->d.bvh_depth > 16 : c->overlap != 0;
        const bool overlap = want_overlap && !c->time_kernels && sc->d.n_lights > 0 && sc->d.n_media == 0 && !piped && (!I->mid_pass || c->overlap == 1);   // (an explicit HK_OVERLAP=1 is obeyed: the tests' small films)
        if (overlap && !c->aux) {
            HIP_TRY(hipStreamCreateWithFlags(&c->aux, hipStreamNonBlocking));
            HIP_TRY(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming));
        }
        bool shadows_in_flight = false;
        // whatever way this function is left while a shadow kernel runs on the second stream, the stream is joined first: the next
        // render's memsets of tickets and counters on `s` must not race with it
        struct AuxJoin {
            hk_ctx* c;
            const bool* in_flight;
            ~AuxJoin() {
                if (*in_flight && c->aux) (void)hipStreamSynchronize(c->aux);
            }
        } aux_join{c, &shadows_in_flight};
        for (int depth = 0; depth < I->p.max_depth; ++depth) {
            if (timed(0, [&] { hk::launch_trace(s, trace_blocks, I->st, sc->d, c->tables, fr, depth, dstats); }) != HK_OK) return fail(HK_ERR_DEVICE, "event record failed");
            c->trace_launches++;
            if (shadows_in_flight) {
                HIP_TRY(hipStreamWaitEvent(s, c->ev_join, 0));
                shadows_in_flight = false;
            }
            int first_kind = 1;
            if (sc->d.n_media > 0) {
                if (timed(4, [&] { hk::launch_medium(s, c->n_cu, I->st, sc->d, c->tables, fr, sob, depth, dstats); }) != HK_OK) return fail(HK_ERR_DEVICE, "event record failed");
                c->media_launches += 2;   // k_track + k_scatter
                first_kind = 0;
            }
            if (sc->d.has_escape_lights) {
                if (lists({{depth, Q_ESCAPED}}, true, depth) != HK_OK) return fail(HK_ERR_DEVICE, "event record failed");
                if (timed(3, [&] { hk::launch_escaped(s, light_blocks, I->st, sc->d, c->tables, fr, depth); }) != HK_OK) return fail(HK_ERR_DEVICE, "event record failed");
            } else if (lists({}, true, depth) != HK_OK)
                return fail(HK_ERR_DEVICE, "event record failed");
            // scenes with a deep light BVH: the next-event light of every shading vertex of this depth, chosen by a kernel of its own
            // (per-lane descent with refill) — part of the shade class
            if (sc->d.n_lights > 0 && hk::preselect_lights(sc->d, I->st)) {
                if (timed(5, [&] { hk::launch_light_select(s, shade_blocks, I->st, sc->d, c->tables, fr, sob, depth, sc->kinds_mask, dstats); }) != HK_OK) return fail(HK_ERR_DEVICE, "event record failed");
                c->select_launches++;
            }
            for (int kind = 0; kind < HK_MAX_KINDS; ++kind)
                if (sc->kinds_mask & (1u << kind)) {
                    if (timed(2, [&] { hk::launch_shade(s, shade_blocks, kind, I->st, sc->d, c->tables, fr, sob, depth, first_kind, dstats); }) != HK_OK) return fail(HK_ERR_DEVICE, "event record failed");
                    first_kind = 0;
                    c->shade_launches++;
                }
            if (lists({{depth, Q_SHADOW}, {depth + 1, Q_RAY}}) != HK_OK) return fail(HK_ERR_DEVICE, "event record failed");
            if (sc->d.n_lights > 0) {
                if (overlap) {
                    HIP_TRY(hipEventRecord(c->ev_fork, s));
                    HIP_TRY(hipStreamWaitEvent(c->aux, c->ev_fork, 0));
                    hk::launch_shadow(c->aux, trace_blocks, I->st, sc->d, c->tables, fr, depth, dstats + c->stat_rows / 2);
                    HIP_TRY(hipEventRecord(c->ev_join, c->aux));
                    shadows_in_flight = true;
                } else
                    if (timed(1, [&] { hk::launch_shadow(s, trace_blocks, I->st, sc->d, c->tables, fr, depth, dstats); }) != HK_OK) return fail(HK_ERR_DEVICE, "event record failed");
                c->shadow_launches++;
            }
        }
        if (shadows_in_flight) {
            HIP_TRY(hipStreamWaitEvent(s, c->ev_join, 0));
            shadows_in_flight = false;
        }
        }   // !fused
        if (piped && c->film_chain) HIP_TRY(hipStreamWaitEvent(s, c->ev_film, 0));   // the film sums in call order
        if (!film_inline && timed(3, [&] { hk::launch_film(s, I->st, fr, c->tables, film->accum, film->f64); }) != HK_OK) return fail(HK_ERR_DEVICE, "event record failed");
        if (piped) {
            HIP_TRY(hipEventRecord(c->ev_film, s));
            c->film_chain = true;
        }
        done += k;
    }
    HIP_TRY(hipGetLastError());
    if (piped) {
        HIP_TRY(hipEventRecord(c->lanes[lane_idx].done, s));
        c->lanes_dirty = true;
    } else
        HIP_TRY(hipEventRecord(c->ev_end, s));
    return HK_OK;
}

extern "C" int32_t hk_film_fill_aux(hk_ctx* c, hk_scene* sc, const hk_camera* cam, int32_t w, int32_t h, int32_t has_infinite_lights, float* albedo, float* normal, float* depth) {
    if (!c || !sc || !cam || !albedo || !normal || !depth || w <= 0 || h <= 0) return fail(HK_ERR_INVALID, "bad argument");
    HIP_TRY(hipSetDevice(c->device));
    DevBuf da, dn, dd;
    const size_t n = (size_t)w * h;
    HIP_TRY(da.alloc(n * 12));
    HIP_TRY(dn.alloc(n * 12));
    HIP_TRY(dd.alloc(n * 4));
    hk::launch_aux(c->stream, sc->d, make_camera(*cam), h, w, has_infinite_lights ? 1e30f : INFINITY, da.as<float>(), dn.as<float>(), dd.as<float>());
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(albedo, da.p, n * 12, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(normal, dn.p, n * 12, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(depth, dd.p, n * 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return HK_OK;
}

extern "C" int32_t hk_sync(hk_ctx* c) {
    if (!c) return fail(HK_ERR_INVALID, "null ctx");
    HIP_TRY(hipSetDevice(c->device));
    if (int e = join_lanes(c)) return e;
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (c->aux) HIP_TRY(hipStreamSynchronize(c->aux));   // a render that failed half-way may have left a shadow kernel on the second stream
    return HK_OK;
}
extern "C" int32_t hk_stats_enable_counters(hk_ctx* c, int32_t flags) {
    if (!c) return fail(HK_ERR_INVALID, "null ctx");
    if (int e = join_lanes(c)) return e;   // (noted small calls are rendered under the flags they were made with)
    c->count_nodes = (flags & 1) ? 1 : 0;
    c->time_kernels = (flags & 2) ? 1 : 0;
    return HK_OK;
}
extern "C" int32_t hk_stats_reset(hk_ctx* c) {
    if (!c) return fail(HK_ERR_INVALID, "null ctx");
    HIP_TRY(hipSetDevice(c->device));
    if (int e = join_lanes(c)) return e;
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipMemset(c->stats.p, 0, (size_t)c->stat_rows * sizeof(DStats)));
    for (auto& l : c->lanes)
        if (l.stats.p) HIP_TRY(hipMemset(l.stats.p, 0, (size_t)c->stat_rows * sizeof(DStats)));
    for (auto& e : c->trace_events) {
        c->event_pool.push_back(e.first);
        c->event_pool.push_back(e.second);
    }
    c->trace_events.clear();
    for (auto& v : c->class_events) {
        for (auto& e : v) {
            c->event_pool.push_back(e.first);
            c->event_pool.push_back(e.second);
        }
        v.clear();
    }
    c->seconds_trace = c->seconds_total = 0.0;
    c->trace_launches = c->shadow_launches = c->shade_launches = c->media_launches = c->select_launches = 0;
    c->fused_passes = 0;
    c->have_span = false;
    return HK_OK;
}
extern "C" int32_t hk_stats_get(hk_ctx* c, hk_stats* out) {
    if (!c || !out) return fail(HK_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(c->device));
    if (int e = join_lanes(c)) return e;
    HIP_TRY(hipStreamSynchronize(c->stream));
    DStats h{};
    {
        // the context's own block + the blocks of the lanes that exist (none unless HK_PIPELINE > 1 has ever started them)
        std::vector<const void*> blocks{c->stats.p};
        for (const auto& l : c->lanes)
            if (l.stats.p) blocks.push_back(l.stats.p);
        std::vector<DStats> rows((size_t)c->stat_rows * blocks.size());
        for (size_t b = 0; b < blocks.size(); ++b)
            HIP_TRY(hipMemcpy(rows.data() + b * (size_t)c->stat_rows, blocks[b], (size_t)c->stat_rows * sizeof(DStats), hipMemcpyDeviceToHost));
        for (const DStats& r : rows) {
            h.rays_closest += r.rays_closest;
            h.rays_shadow += r.rays_shadow;
            h.nodes += r.nodes;
            h.tris += r.tris;
            h.hits += r.hits;
            h.vertices += r.vertices;
            h.collisions += r.collisions;
            h.light_nodes += r.light_nodes;
            h.sh_nodes += r.sh_nodes;
            h.sh_tris += r.sh_tris;
            h.sh_collisions += r.sh_collisions;
            h.nvdb_collisions += r.nvdb_collisions;
            h.sh_nvdb_collisions += r.sh_nvdb_collisions;
            h.dda_steps += r.dda_steps;
            h.sh_dda_steps += r.sh_dda_steps;
            h.scatter_vertices += r.scatter_vertices;
            h.sc_light_nodes += r.sc_light_nodes;
#ifdef HK_DEBUG_UTIL
            for (int k = 0; k < 32; ++k) h.dbg[k] += r.dbg[k];
#endif
        }
#ifdef HK_DEBUG_UTIL
        for (int k = 0; k < 16; ++k)
            if (h.dbg[2 * k]) std::fprintf(stderr, "HK_DEBUG_UTIL[%d]: %.3f of %llu lane-slots\n", k, (double)h.dbg[2 * k + 1] / (double)h.dbg[2 * k], h.dbg[2 * k]);
#endif
    }
    double tr = 0.0;
    for (auto& e : c->trace_events) {
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, e.first, e.second) == hipSuccess) tr += ms * 1e-3;
    }
    double total = 0.0;
    if (c->have_span) {
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, c->ev_begin, c->ev_end) == hipSuccess) total = ms * 1e-3;
    }
    std::memset(out, 0, sizeof *out);
    out->rays_closest = h.rays_closest;
    out->rays_shadow = h.rays_shadow;
    out->bvh_nodes_visited = h.nodes;
    out->tris_tested = h.tris;
    out->hits_accepted = h.hits;
    out->path_vertices = h.vertices;
    out->medium_collisions = h.collisions + h.sh_collisions;
    out->track_collisions = h.collisions;
    out->shadow_collisions = h.sh_collisions;
    out->track_dda_steps = h.dda_steps;
    out->shadow_dda_steps = h.sh_dda_steps;
    out->scatter_vertices = h.scatter_vertices;
    out->light_bvh_nodes = h.light_nodes + h.sc_light_nodes;
    out->seconds_trace = tr;
    out->seconds_total = total;
    out->trace_launches = c->trace_launches;
    out->trace_nodes = h.nodes;
    out->trace_tris = h.tris;
    out->shadow_nodes = h.sh_nodes;
    out->shadow_tris = h.sh_tris;
    out->bvh_nodes_visited = h.nodes + h.sh_nodes;
    out->tris_tested = h.tris + h.sh_tris;
    out->shadow_launches = c->shadow_launches;
    out->shade_launches = c->shade_launches;
    out->media_launches = c->media_launches;
    double cls[6] = {0, 0, 0, 0, 0, 0};
    for (int k = 1; k < 6; ++k)
        for (auto& e : c->class_events[k]) {
            float ms = 0.0f;
            if (hipEventElapsedTime(&ms, e.first, e.second) == hipSuccess) cls[k] += ms * 1e-3;
        }
    out->seconds_shadow = cls[1];
    out->seconds_shade = cls[2] + cls[5];   // (the light selection of a deep light BVH is part of K9: inside the shade class, and on its own below)
    out->seconds_other = cls[3];
    out->seconds_media = cls[4];
    out->seconds_select = cls[5];
    out->select_launches = c->select_launches;
    out->fused_passes = c->fused_passes;
    {   // SURVEY 8(d) algorithmic bytes over the counted units
        const uint64_t hits_closest = h.hits < h.rays_closest ? h.hits : h.rays_closest;   // shading attributes are fetched once per accepted closest hit
        out->bytes_algorithmic_trace = h.rays_closest * (32 + 16) + 64 * h.nodes + 36 * h.tris + 96 * hits_closest;
        out->bytes_algorithmic_shadow = h.rays_shadow * (32 + 16) + 64 * h.sh_nodes + 36 * h.sh_tris;
        out->bytes_algorithmic_shade = h.vertices * (2 * 104 + 64 + 96) + 60 * h.light_nodes;
        // media: per collision 8 taps x 4 B + 4 B majorant = 36 B on a dense grid, 84 B through a NanoVDB tree (+ 8 B x 3
        // levels x 2 leaves); per majorant cell entered (DDA step) 4 B; the delta-tracking kernel reads and rewrites the path state
        // (2 x 104 B) once per tracked ray = per entry of the medium queue, which is what `track_rays` counts; a scattering vertex
        // (K5 + K6) is a path vertex without a material record.
        // The kernels count the collisions of NanoVDB scenes apart (DStats::nvdb_collisions), so a context that renders grid and NanoVDB
        // scenes in one statistics window charges each its own price.
        out->bytes_algorithmic_media = h.collisions * 36 + h.nvdb_collisions * (84 - 36) + 4 * h.dda_steps + h.scatter_vertices * (2 * 104 + 96) + 60 * h.sc_light_nodes;
        out->bytes_algorithmic_shadow += h.sh_collisions * 36 + h.sh_nvdb_collisions * (84 - 36) + 4 * h.sh_dda_steps;
    }
    return HK_OK;
}

// ---- sub-kernel entry points --------------------------------------------------------------------------
namespace {
struct Tmp {
    std::vector<void*> ptrs;
    ~Tmp() {
        for (void* p : ptrs) (void)hipFree(p);
    }
    template <class T>
    T* up(const T* src, size_t n) {
        void* p = nullptr;
        if (hipMalloc(&p, (n ? n : 1) * sizeof(T)) != hipSuccess) return nullptr;
        ptrs.push_back(p);
        if (src && n) (void)hipMemcpy(p, src, n * sizeof(T), hipMemcpyHostToDevice);
        return (T*)p;
    }
};
}  // namespace

extern "C" int32_t hk_trace_closest(hk_ctx* c, hk_scene* sc, int32_t n, const float* o3, const float* d3, const float* tmax, float* out_t, int32_t* out_prim,
                                    float* out_uv2) {
    if (!c || !sc || n < 0) return fail(HK_ERR_INVALID, "bad argument");
    HIP_TRY(hipSetDevice(c->device));
    KnobScope knobs(&c->knobs);
    Tmp t;
    float *o = t.up(o3, 3 * (size_t)n), *d = t.up(d3, 3 * (size_t)n), *tm = t.up(tmax, n);
    float* ot = t.up<float>(nullptr, n);
    int* op = t.up<int>(nullptr, n);
    float* ouv = t.up<float>(nullptr, 2 * (size_t)n);
    if (!o || !d || !tm || !ot || !op || !ouv) return fail(HK_ERR_DEVICE, "hipMalloc failed");
    hk::launch_test_trace(c->stream, sc->d, n, o, d, tm, ot, op, ouv);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipMemcpy(out_t, ot, n * 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(out_prim, op, n * 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(out_uv2, ouv, 2 * (size_t)n * 4, hipMemcpyDeviceToHost));
    return HK_OK;
}
extern "C" int32_t hk_test_sobol(hk_ctx* c, int32_t width, int32_t height, int32_t spp, uint32_t seed, int32_t n, const int32_t* px, const int32_t* py,
                                 const int32_t* sample_idx, const int32_t* dim, float* out_1d, float* out_2d) {
    if (!c || !c->have_tables) return fail(HK_ERR_INVALID, "tables not set");
    HIP_TRY(hipSetDevice(c->device));
    hk_integrator_params p{};
    p.samples_per_pixel = spp;
    p.sampler_seed = seed;
    DSobol sob;
    sob.log2_spp = ceil_log2(spp < 1 ? 1 : spp);
    sob.n_base4_digits = ceil_log2(width > height ? width : height) + (sob.log2_spp + 1) / 2;
    sob.seed = seed;
    sob.width = width;
    Tmp t;
    int *a = t.up(px, n), *b = t.up(py, n), *s = t.up(sample_idx, n), *dm = t.up(dim, n);
    float *o1 = t.up<float>(nullptr, n), *o2 = t.up<float>(nullptr, 2 * (size_t)n);
    hk::launch_test_sobol(c->stream, c->tables, sob, n, a, b, s, dm, o1, o2);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipMemcpy(out_1d, o1, n * 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(out_2d, o2, 2 * (size_t)n * 4, hipMemcpyDeviceToHost));
    return HK_OK;
}
extern "C" int32_t hk_test_camera(hk_ctx* c, hk_integrator* I, const hk_camera* cam, int32_t width, int32_t height, int32_t n, const int32_t* px, const int32_t* py,
                                  const int32_t* sample_idx, float* out15) {
    if (!c || !I || !cam) return fail(HK_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(c->device));
    DSobol sob = make_sobol(I->p, width, height);
    DCamera dc = make_camera(*cam);
    Tmp t;
    int *a = t.up(px, n), *b = t.up(py, n), *s = t.up(sample_idx, n);
    float* o = t.up<float>(nullptr, 15 * (size_t)n);
    hk::launch_test_camera(c->stream, c->tables, I->filter, dc, sob, height, n, a, b, s, o);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipMemcpy(out15, o, 15 * (size_t)n * 4, hipMemcpyDeviceToHost));
    return HK_OK;
}
extern "C" int32_t hk_test_uplift(hk_ctx* c, int32_t mode, int32_t n, const float* rgb, const float* lambda, float* out) {
    if (!c || !c->have_tables) return fail(HK_ERR_INVALID, "tables not set");
    HIP_TRY(hipSetDevice(c->device));
    Tmp t;
    float *r = t.up(rgb, 3 * (size_t)n), *l = t.up(lambda, 4 * (size_t)n), *o = t.up<float>(nullptr, 4 * (size_t)n);
    hk::launch_test_uplift(c->stream, c->tables, mode, n, r, l, o);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipMemcpy(out, o, 4 * (size_t)n * 4, hipMemcpyDeviceToHost));
    return HK_OK;
}
extern "C" int32_t hk_test_light(hk_ctx* c, hk_scene* sc, int32_t mode, int32_t light_idx_1based, int32_t n, const float* p3, const float* in3, const float* lambda,
                                 float* out) {
    if (!c || !sc || !p3 || !in3 || !lambda || !out) return fail(HK_ERR_INVALID, "null argument");
    if (mode == 0 && (light_idx_1based < 1 || light_idx_1based > sc->d.n_lights)) return fail(HK_ERR_INVALID, "light index out of range");
    HIP_TRY(hipSetDevice(c->device));
    Tmp t;
    float *dp = t.up(p3, 3 * (size_t)n), *di = t.up(in3, 3 * (size_t)n), *dl = t.up(lambda, 4 * (size_t)n), *o = t.up<float>(nullptr, 12 * (size_t)n);
    hk::launch_test_light(c->stream, sc->d, c->tables, mode, light_idx_1based, n, dp, di, dl, o);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipMemcpy(out, o, 12 * (size_t)n * 4, hipMemcpyDeviceToHost));
    return HK_OK;
}
extern "C" int32_t hk_test_bsdf(hk_ctx* c, hk_scene* sc, int32_t mode, int32_t mat_idx, int32_t regularize, int32_t n, const float* wo, const float* wi, const float* ns,
                                const float* lambda, const float* u, const float* uc, float* out) {
    if (!c || !sc || !wo || !wi || !ns || !lambda || !u || !uc || !out) return fail(HK_ERR_INVALID, "null argument");
    if (mat_idx < 0 || mat_idx >= sc->n_materials) return fail(HK_ERR_INVALID, "material index out of range");
    HIP_TRY(hipSetDevice(c->device));
    Tmp t;
    float *dwo = t.up(wo, 3 * (size_t)n), *dwi = t.up(wi, 3 * (size_t)n), *dns = t.up(ns, 3 * (size_t)n), *dl = t.up(lambda, 4 * (size_t)n);
    float *du = t.up(u, 2 * (size_t)n), *duc = t.up(uc, (size_t)n), *o = t.up<float>(nullptr, 10 * (size_t)n);
    hk::launch_test_bsdf(c->stream, sc->d, c->tables, mode, mat_idx, regularize, n, dwo, dwi, dns, dl, du, duc, o);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipMemcpy(out, o, 10 * (size_t)n * 4, hipMemcpyDeviceToHost));
    return HK_OK;
}
extern "C" int32_t hk_test_light_bvh(hk_ctx* c, hk_scene* sc, int32_t n, const float* p3, const float* n3, const float* u, int32_t* out_light, float* out_pmf,
                                     const int32_t* query_light, float* out_query_pmf) {
    if (!c || !sc) return fail(HK_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(c->device));
    Tmp t;
    float *p = t.up(p3, 3 * (size_t)n), *nn = t.up(n3, 3 * (size_t)n), *uu = t.up(u, n);
    int* ol = t.up<int>(nullptr, n);
    float* op = t.up<float>(nullptr, n);
    int* q = query_light ? t.up(query_light, n) : nullptr;
    float* oq = t.up<float>(nullptr, n);
    hk::launch_test_light_bvh(c->stream, sc->d, n, p, nn, uu, ol, op, q, oq);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipMemcpy(out_light, ol, n * 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(out_pmf, op, n * 4, hipMemcpyDeviceToHost));
    if (query_light && out_query_pmf) HIP_TRY(hipMemcpy(out_query_pmf, oq, n * 4, hipMemcpyDeviceToHost));
    return HK_OK;
}

extern "C" int32_t hk_test_mix(hk_ctx* c, hk_scene* sc, int32_t mat_idx, int32_t n, const float* p3, const float* wo3, const float* uv2, int32_t* out_mat) {
    if (!c || !sc || !p3 || !wo3 || !uv2 || !out_mat) return fail(HK_ERR_INVALID, "null argument");
    if (mat_idx < 0 || mat_idx >= sc->n_materials) return fail(HK_ERR_INVALID, "material index out of range");
    HIP_TRY(hipSetDevice(c->device));
    Tmp t;
    float *dp = t.up(p3, 3 * (size_t)n), *dw = t.up(wo3, 3 * (size_t)n), *du = t.up(uv2, 2 * (size_t)n);
    int* o = t.up<int>(nullptr, (size_t)n);
    hk::launch_test_mix(c->stream, sc->d, mat_idx, n, dp, dw, du, o);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipMemcpy(out_mat, o, (size_t)n * 4, hipMemcpyDeviceToHost));
    return HK_OK;
}
extern "C" int32_t hk_test_medium(hk_ctx* c, hk_scene* sc, int32_t mode, int32_t medium_idx, int32_t n, const float* a3, const float* b3, const float* tmax,
                                  const float* lambda, float* out) {
    if (!c || !sc || !a3 || !lambda || !out || (mode >= 1 && (!b3 || !tmax))) return fail(HK_ERR_INVALID, "null argument");
    if (mode < 0 || mode > 2) return fail(HK_ERR_INVALID, "mode must be 0 (sample_point), 1 (majorant segments) or 2 (majorant segments with the zero-cell fast-forward)");
    if (medium_idx < 0 || medium_idx >= sc->d.n_media) return fail(HK_ERR_INVALID, "medium index out of range");
    HIP_TRY(hipSetDevice(c->device));
    const size_t stride = mode == 0 ? 13 : (size_t)hk::test_majorant_stride();
    Tmp t;
    float *da = t.up(a3, 3 * (size_t)n), *db = b3 ? t.up(b3, 3 * (size_t)n) : nullptr, *dt = tmax ? t.up(tmax, (size_t)n) : nullptr, *dl = t.up(lambda, 4 * (size_t)n);
    float* o = t.up<float>(nullptr, stride * (size_t)n);
    hk::launch_test_medium(c->stream, sc->d, c->tables, mode, medium_idx, n, da, db, dt, dl, o);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipMemcpy(out, o, stride * (size_t)n * 4, hipMemcpyDeviceToHost));
    return HK_OK;
}
extern "C" int32_t hk_test_trace_lean(hk_ctx* c, hk_scene* sc, int32_t anyhit, int32_t n, const float* o3, const float* d3, const float* tmax, float* out_t, int32_t* out_prim,
                                      float* out_uv2) {
    if (!c || !sc || !o3 || !d3 || !tmax || !out_t || !out_prim || !out_uv2) return fail(HK_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(c->device));
    KnobScope knobs(&c->knobs);
    Tmp t;
    float *o = t.up(o3, 3 * (size_t)n), *d = t.up(d3, 3 * (size_t)n), *tm = t.up(tmax, (size_t)n);
    float *ot = t.up<float>(nullptr, (size_t)n), *ouv = t.up<float>(nullptr, 2 * (size_t)n);
    int* op = t.up<int>(nullptr, (size_t)n);
    hk::launch_test_trace_lean(c->stream, c->n_cu, sc->d, anyhit, n, o, d, tm, ot, op, ouv);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipMemcpy(out_t, ot, (size_t)n * 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(out_prim, op, (size_t)n * 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(out_uv2, ouv, 2 * (size_t)n * 4, hipMemcpyDeviceToHost));
    return HK_OK;
}

// ---------------------------------------------------------------------------------------------------------------------------
// Multi-GPU behind the C-ABI: the ONE exchange step of the path is a sum-reduce of the film accumulators
// [pixel_rgb 3N | pixel_weight_sum N] (volpath.jl:364-373, volpath-state.jl:122-131) to one device after the last sample.  It is done
// here with RCCL directly (ncclReduce over xGMI), so a caller without torch — the Julia shim with `devices = 0:7` — shards a frame
// over the GPUs of a node: one hk_ctx per device, hk_render on each (asynchronous, each on its context's stream), hk_film_reduce,
// hk_film_read_rgb on the root.  librccl is loaded on first use (dlopen): single-GPU users never touch it.
// ---------------------------------------------------------------------------------------------------------------------------
namespace {
struct Rccl {
    void* lib = nullptr;
    // the few entry points used, with the prototypes of <rccl/rccl.h> (ncclResult_t = int, ncclComm_t = opaque pointer,
    // ncclUniqueId = 128 opaque bytes passed by value, ncclDataType_t: ncclFloat32 = 7, ncclFloat64 = 8, ncclRedOp_t: ncclSum = 0)
    struct UniqueId {
        char internal[128];
    };
    int (*GetUniqueId)(UniqueId*) = nullptr;
    int (*CommInitRank)(void**, int, UniqueId, int) = nullptr;
    int (*CommInitAll)(void**, int, const int*) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    int (*Reduce)(const void*, void*, size_t, int, int, int, void*, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    std::string error;
};
Rccl& rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : names) {
            r.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
            if (r.lib) break;
        }
        if (!r.lib) {
            r.error = std::string("cannot load librccl: ") + (dlerror() ? dlerror() : "?");
            return;
        }
        auto sym = [&](const char* name) {
            void* p = dlsym(r.lib, name);
            if (!p && r.error.empty()) r.error = std::string("librccl lacks ") + name;
            return p;
        };
        r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
        r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
        r.CommInitAll = reinterpret_cast<decltype(r.CommInitAll)>(sym("ncclCommInitAll"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
        r.Reduce = reinterpret_cast<decltype(r.Reduce)>(sym("ncclReduce"));
        r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(sym("ncclGroupStart"));
        r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(sym("ncclGroupEnd"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
    });
    return r;
}
int rccl_fail(const char* what, int code) {
    Rccl& r = rccl();
    return fail(HK_ERR_DEVICE, std::string(what) + ": " + (r.GetErrorString ? r.GetErrorString(code) : "RCCL error"));
}
#define RCCL_TRY(expr)                              \
    do {                                            \
        int rc_ = (expr);                           \
        if (rc_ != 0) return rccl_fail(#expr, rc_); \
    } while (0)
}  // namespace

struct hk_comm {
    std::vector<hk_ctx*> ctxs;     // local ranks of this process (1 in the one-process-per-GPU layout)
    std::vector<void*> comms;      // ncclComm_t per local rank
    int world = 1;
};

extern "C" int32_t hk_comm_create(hk_ctx* const* ctxs, int32_t n, hk_comm** out) {
    if (!ctxs || n < 1 || !out) return fail(HK_ERR_INVALID, "bad argument");
    for (int i = 0; i < n; ++i) {
        if (!ctxs[i]) return fail(HK_ERR_INVALID, "null context");
        for (int j = 0; j < i; ++j)
            if (ctxs[j]->device == ctxs[i]->device) return fail(HK_ERR_INVALID, "two contexts on the same device in one communicator");
    }
    Rccl& r = rccl();
    if (!r.error.empty()) return fail(HK_ERR_UNSUPPORTED, r.error);
    std::unique_ptr<hk_comm> c(new hk_comm());
    c->ctxs.assign(ctxs, ctxs + n);
    c->comms.assign(n, nullptr);
    c->world = n;
    std::vector<int> devs(n);
    for (int i = 0; i < n; ++i) devs[i] = ctxs[i]->device;
    RCCL_TRY(r.CommInitAll(c->comms.data(), n, devs.data()));
    *out = c.release();
    return HK_OK;
}
extern "C" int32_t hk_comm_unique_id(uint8_t* id_out128) {
    if (!id_out128) return fail(HK_ERR_INVALID, "null argument");
    Rccl& r = rccl();
    if (!r.error.empty()) return fail(HK_ERR_UNSUPPORTED, r.error);
    Rccl::UniqueId id;
    RCCL_TRY(r.GetUniqueId(&id));
    std::memcpy(id_out128, id.internal, 128);
    return HK_OK;
}
extern "C" int32_t hk_comm_create_rank(hk_ctx* ctx, const uint8_t* id128, int32_t rank, int32_t world, hk_comm** out) {
    if (!ctx || !id128 || !out || world < 1 || rank < 0 || rank >= world) return fail(HK_ERR_INVALID, "bad argument");
    Rccl& r = rccl();
    if (!r.error.empty()) return fail(HK_ERR_UNSUPPORTED, r.error);
    HIP_TRY(hipSetDevice(ctx->device));
    Rccl::UniqueId id;
    std::memcpy(id.internal, id128, 128);
    std::unique_ptr<hk_comm> c(new hk_comm());
    c->ctxs.assign(1, ctx);
    c->comms.assign(1, nullptr);
    c->world = world;
    RCCL_TRY(r.CommInitRank(&c->comms[0], world, id, rank));
    *out = c.release();
    return HK_OK;
}
extern "C" int32_t hk_comm_destroy(hk_comm* c) {
    if (!c) return HK_OK;
    Rccl& r = rccl();
    for (size_t i = 0; i < c->comms.size(); ++i)
        if (c->comms[i] && r.CommDestroy) {
            (void)hipSetDevice(c->ctxs[i]->device);
            (void)hipStreamSynchronize(c->ctxs[i]->stream);
            (void)r.CommDestroy(c->comms[i]);
        }
    delete c;
    return HK_OK;
}
extern "C" int32_t hk_film_reduce(hk_comm* c, hk_film* const* films, int32_t n_films, int32_t root) {
    if (!c || !films || n_films != (int32_t)c->comms.size()) return fail(HK_ERR_INVALID, "one film per local rank of the communicator is required");
    if (root < 0 || root >= c->world) return fail(HK_ERR_INVALID, "root out of range");
    for (int i = 0; i < n_films; ++i) {
        if (!films[i] || films[i]->ctx != c->ctxs[i]) return fail(HK_ERR_INVALID, "film i must live on the communicator's context i");
        if (films[i]->width != films[0]->width || films[i]->height != films[0]->height || films[i]->f64 != films[0]->f64)
            return fail(HK_ERR_INVALID, "films differ in size or accumulation type");
    }
    Rccl& r = rccl();
    if (!r.error.empty()) return fail(HK_ERR_UNSUPPORTED, r.error);
    const size_t count = (size_t)4 * films[0]->width * films[0]->height;
    const int dtype = films[0]->f64 ? 8 : 7;   // ncclFloat64 : ncclFloat32
    if (n_films > 1) RCCL_TRY(r.GroupStart());
    // once the group is open it is ALWAYS closed: a return between GroupStart and GroupEnd would leave the process-wide RCCL group
    // open and every later RCCL call queued behind it.  The first error is kept, the loop is left, GroupEnd runs, then the error returns.
    int status = HK_OK;
    for (int i = 0; i < n_films && status == HK_OK; ++i) {
        const hipError_t he = hipSetDevice(c->ctxs[i]->device);
        if (he != hipSuccess) {
            status = fail(HK_ERR_DEVICE, std::string("hipSetDevice: ") + hipGetErrorString(he));
            break;
        }
        // in place: the root's accumulators receive the sum; ordered after the renders already enqueued on the context's stream
        if (join_lanes(c->ctxs[i]) != HK_OK) {
            status = HK_ERR_DEVICE;
            break;
        }
        const int rc = r.Reduce(films[i]->accum, films[i]->accum, count, dtype, 0 /* ncclSum */, root, c->comms[i], c->ctxs[i]->stream);
        if (rc != 0) status = rccl_fail("ncclReduce", rc);
    }
    if (n_films > 1) {
        const int rc = r.GroupEnd();
        if (rc != 0 && status == HK_OK) status = rccl_fail("ncclGroupEnd", rc);
    }
    return status;
}

// hk_types.h — device-resident records of the MI355X VolPath path (shared by host upload code and kernels).
//
// HBM layout (DESIGN.md "Data layout"):
//   BVH2 nodes        64 B  = 4 x float4 : both child AABBs + 2 child refs        (one 64-B line per visit)
//   leaf triangles    48 B  = 3 x float4 : v0|prim id, e1|flags, e2               (leaf order)
//   shading triangles 36 B positions + 36 B normals + 24 B uvs + 12 B meta        (original prim order)
//   path state        SoA float4 arrays in QUEUE ORDER, two generations ping-ponged by depth (DPathGen); film inputs per path slot
//   queues            the generation arrays are the ray queue; per-kind / escaped / medium queues hold uint32 generation indices;
//                     shadow records are stored in shadow-queue order; device-side counters (no host readback inside a frame)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// Run-time knobs (hk_api.cpp): the value of HK_<name> in the table of the context whose entry point runs on this thread, or nullptr.
// The environment is read once, by hk_ctx_create; hk_ctx_set_option changes a knob afterwards.
namespace hk {
const char* knob(const char* name);
}

#define HK_TICKET_COLS 16       // kernels per bounce that draw segment tickets (5 + HK_MAX_KINDS)
#define HK_TICKET_WAYS 64       // counters one ticket is split into: one per lane of a wave
#define HK_TICKET_STRIDE 64     // ints between two counters of a ticket: one 256-byte line each
#define HK_MAX_KINDS 11          // HK_MAT_* count (Mix is resolved before queueing)
#ifndef HK_LDS_STACK
#define HK_LDS_STACK 32          // per-lane traversal stack entries kept in LDS (builder bounds the depth)
#endif
#define HK_TRACE_BLOCK 256

// child reference: >= 0 inner node index; < 0 leaf: ~ref = (first_tri << 3) | (count - 1), count <= 4 (<=8 encodable)
struct DNode {
    float4 a;  // lo0.x hi0.x lo0.y hi0.y   (the (lo, hi) pair of an axis in adjacent words: one packed fma per pair)
    float4 b;  // lo0.z hi0.z lo1.x hi1.x
    float4 c;  // lo1.y hi1.y lo1.z hi1.z
    int c0, c1;
    int pad0, pad1;
};

#define HK_MAT_EMISSIVE_BIT 0x40000000   // DPathState::mat_id: the hit triangle carries an area light (set by the trace kernels)
#define HK_TRI_OPAQUE 1u  // e1.w bit: surface is neither a medium transition nor alpha-tested

// sigmoid-polynomial spectrum with optional scale (c0,c1,c2,scale)
struct DSpectrumParam {
    float4 coef;     // baked (constant colour): sigmoid coefficients + scale
    float rgba[4];   // raw value (needed when tex >= 0 is blended, or for alpha)
    int tex;         // -1 constant
    int pad[3];
};

struct DMaterial {
    int kind, flags;
    int i[4];
    int spectrum[2];
    DSpectrumParam rgb[4];
    float f[8];
    int ftex[8];
    uint32_t mix_key[4];
};

struct DTexture {
    const float* data;
    int width, height, channels, pad;
};

struct DPLSpectrum {
    const float* lambdas;
    const float* values;
    int n, pad;
};

// 128-byte light record
struct DLight {
    int kind;
    int flags;        // bit0 two_sided, bit1 illuminant (multiply by D65)
    float scale;      // light.scale
    float area;
    float4 coef;      // baked spectrum: sigmoid coefficients + scale (2m for illuminants; Le*scale for area lights)
    float p[3];       // position | direction
    float cos_total_width;
    float cos_falloff_start;
    float normal[3];
    float v[9];       // area-light vertices | spot world_to_light 3x3
    float uv[6];
    int Le_tex;
    float Le_rgba[4]; // raw Le (textured emission path)
    int pad;
};

// EnvironmentMap + Distribution2D (textures/environment_map.jl:9-45; sampler/sampling.jl:179-262), Julia layouts kept
struct DEnvMap {
    const float4* data;            // texel (y, x) at [y + height * x]
    const float* cond_func;        // [nu, nv]: (u, v) at [u + nu * v]
    const float* cond_cdf;         // [nu + 1, nv]
    const float* cond_func_int;    // [nv]
    const float* marg_func;        // [nv]
    const float* marg_cdf;         // [nv + 1]
    float marg_func_int;
    int width, height, nu, nv;
    float rot[9];                  // row-major Mat3f
};

// Light-BVH node as the device walks it, 64 B.  The quantities of node_importance that depend on the node alone are evaluated
// once on the host with the same binary32 operations, in the same order, as the per-visit code they replace
// (light-bounds.jl:96-109, bvh-light-sampler.jl:177-182): centre = (bmin + bmax) * 0.5, half_diag = norm(bmax - bmin) * 0.5,
// r2 = |bmax - centre|^2, sin_o = sqrt(max(0, 1 - cos_o^2)).
struct DLightNode {
    float centre[3];
    float half_diag;
    float r2;
    float w[3];
    float phi, cos_o, cos_e, sin_o;
    uint32_t bits;              // bit0 two_sided, bit1 is_leaf
    uint32_t child1_or_light;   // leaf: light index, 1-based; inner node: ENTRY of its child 0 — child 1 is the next entry, the pair shares one 128-B line (sibling-pair order, hk_api.cpp)
    uint32_t pad[2];
};

struct DMedium {
    int kind;
    float g;
    float4 sigma_a, sigma_s, Le;   // baked uplift_rgb_unbounded coefficients (c0,c1,c2,scale)
    float bmin[3], bmax[3];
    float r2m[12];                 // render_to_medium rows 0..2 (affine)
    int res[3];
    const float* density;          // Grid: [nx,ny,nz] x fastest
    const float4* rgb_a;           // RGBGrid: sigma_a / sigma_s / Le voxels (null = absent)
    const float4* rgb_s;
    const float4* rgb_Le;
    float sigma_scale, Le_scale;
    int mres[3];
    const float* majorant;         // x + rx*(y + ry*z)
    const uint32_t* maj_zero;      // bit c set <=> majorant[c] == 0: empty cells are skipped without touching the float grid
    const unsigned char* nvdb;     // NanoVDB bytes (tree part)
    const uint2* nv_blocks;        // flattened tree over the index bbox: {leaf offset (1-based, 0 = constant block), value bits}
    int nvb_min[3], nvb_dim[3];    // block-coordinate origin / extent of nv_blocks ([bx][by][bz], bz fastest)
    const float* nv_bricks;        // optional: every block of the table materialised as a dense brick with a +1 halo (9^3 floats, z fastest): the 8
                                   // taps of a lookup are four 8-byte loads at ONE computed address instead of table entry -> leaf value (null: use nv_blocks)
    float nv_background;           // value of every block outside the table
    long long root_off;            // 1-based like the reference
    int root_table_size;
    float inv_mat[9], vec[3];
};

struct DMediumInterface {
    int material, inside, outside, pad;
};

struct DTriMeta {
    uint32_t mi, prim_index, arealight;
};

// QUANTISED NODES (trees deeper than 16 levels; round 5).  The deep-tree traversal kernels are bound by the number of 16-B loads a
// node step issues per lane (each one a line look-up in the CU's vector cache: DESIGN.md section 5) — a 64-B node is four of them.  A
// DQNode is two: the twelve box planes as 16-bit coordinates on a grid over the scene's bounds (lo rounded down, hi rounded up, one
// more cell of margin), the same child references.  Boxes only grow, so every leaf the float tree reaches is reached; hits come from the
// same triangle tests, and ties on t are broken by triangle index, not by visiting order: results are those of the float tree.
struct DQNode {
    uint32_t w[6];   // (lo | hi << 16) per axis pair, in DNode's order: c0 x, c0 y, c0 z, c1 x, c1 y, c1 z
    int c0, c1;
};

struct DScene {
    const DNode* nodes;
    const DQNode* qnodes;       // null unless the tree is deeper than 16 levels (and HK_QNODES != 0)
    float q_base[3], q_cell[3]; // plane = q_base + q * q_cell
    const float4* leaf_tris;    // 3 float4 per triangle in leaf order
    int root_ref;
    int n_tris;
    int n_nodes;                // inner nodes, breadth-first: the first HK_NODE_CACHE of them are what the lean traversal kernels keep in LDS
    int pad_nodes;
    const float* positions;     // [T][9]
    const float* normals;       // [T][9] or null
    const float* uvs;           // [T][6] or null
    const float* tangents;      // [T][9] or null
    const DTriMeta* meta;
    const DMaterial* materials;
    const DTexture* textures;
    const DPLSpectrum* spectra;
    const DMediumInterface* mis;
    const DLight* lights;
    int n_lights;
    int n_materials;
    const DLightNode* lnodes;
    const uint32_t* bit_trails;
    const int* infinite_lights;
    int num_bvh_lights, num_infinite_lights;
    const DMedium* media;
    int n_media;
    int media_mask;             // bit k set: a medium of kind k (HK_MEDIUM_*) is present
    const DEnvMap* envmaps;
    int n_envmaps;
    int has_escape_lights;      // any ambient / environment light
    int simple_lights;          // no ambient / environment light and no texture of any kind: k_shade<Matte, true> leaves those branches out
    int all_opaque;             // no medium transitions and no alpha-tested surfaces
    int bvh_depth;              // deepest BVH level (bounds the traversal stack)
    int all_grey;               // every medium is a Grid / NanoVDB medium whose sigma_a and sigma_s are flat spectra (the GREY tracking kernels)
    int grey_pool;              // all_grey, ONE medium, majorant grid <= 1024 cells per axis (k_track_pool packs the cell index in one word)
    int grey_bricks;            // all_grey and the one medium is a NanoVDB grid with dense halo bricks (DMedium::nv_bricks): tracking kernels without the tree walk
    // PACKED SHADING RECORDS (round 6): what a shading vertex reads of its triangle — positions 36 B, vertex normals 36 B (NaN: none), uvs 24 B
    // (the defaults when the mesh has none), meta 12 B — as ONE 128-byte, 128-byte-aligned record per triangle (32 floats: p[9] n[9] uv[6]
    // meta[3] pad[5]) instead of four arrays with strides of 36 / 36 / 24 / 12 B: a hit on a random triangle of the 10^6-triangle scene touches
    // one line instead of five to six 64-byte sectors in four places (profiles/pmc_traffic_manylight.json: k_shade moved 2.4 x its algorithmic
    // bytes).  Same values, same arithmetic.  Null (HK_TRI_PACK=0): the separate arrays.
    const float* tri_shade;
};

struct DTables {
    const uint32_t* sobol;      // dims 0 and 1 (2 x 52)
    const float* cie;           // x[471] y[471] z[471]
    const float* rgb2spec_scale;
    const float* rgb2spec_coeffs;
    int rgb2spec_res;
    const float4* rgb2spec_points;   // the same coefficients as float4 grid points (c0, c1, c2, 0) at [((maxc * res + z) * res + y) * res + x]: one load per corner
    int rgb2spec_sorted;             // 1: rgb2spec_scale is non-decreasing (binary search for the z cell is then exact)
};

struct DFilter {
    int type;
    float rx, ry, p1, p2;
    int nx, ny;
    const float* func;            // ny*nx
    const float* marginal_cdf;    // ny+1
    const float* marginal_func;   // ny
    const float* conditional_cdf; // ny*(nx+1)
    float dmin_x, dmin_y, dmax_x, dmax_y;
    float func_integral;
};

struct DCamera {
    float r2c[16];
    float c2w[16];
    float lens_radius, focal_distance, shutter_open, shutter_close;
};

struct DSobol {
    int log2_spp, n_base4_digits;
    uint32_t seed;
    int width;
    // Pixel part of the permuted ZSobol index, per (dimension row, pixel slot): the base-4 digits above the sample bits and
    // their permutations depend on (pixel, dimension) only, so they are tabulated once per film size (10 of the 16 digit
    // hashes of a draw at 800^2).  row = sobol_row(dimension), entry = permuted digits >> log2_spp.  Null = compute in full.
    const uint2* hi_table;     // x = permuted digits above the sample bits (>> log2_spp); y = permutation indices of the two top SAMPLE digits (zsobol_top_perms)
    int hi_rows, hi_stride;   // rows available, entries per row (= n_pixels_padded)
    // The permuted SAMPLE bits (low log2_spp bits of the index) of every sample index one render call draws, per (row, pixel slot):
    // entry ((row * hi_stride + pixel slot) * lo_count + j) belongs to sample index lo_base + j * sample_stride.  With path slots
    // sample-fastest the lanes of a wave read neighbouring entries; a draw is then one 2-byte load instead of a 64-bit hash and a
    // mod-24 per remaining base-4 digit.  Built per (film, first sample, stride, count) when a call renders enough samples.
    const uint16_t* lo_table;
    int lo_rows, lo_count;    // rows tabulated (the first lo_rows of the hi rows), entries per (row, pixel slot)
    int lo_offset;            // per pass: entry index of the pass's sample k = 0
};

// queue ids inside one depth's counter block
enum { Q_RAY = 0, Q_SHADOW = 1, Q_ESCAPED = 2, Q_MEDIUM = 3, Q_SCATTER = 4, Q_MAT0 = 5, Q_COUNT = Q_MAT0 + HK_MAX_KINDS };

// One generation of live-path records IN QUEUE ORDER.  Entry p = segment * wave_cap + position: the paths of wave segment `gw`
// that are alive at depth d occupy the dense prefix [gw * wave_cap, gw * wave_cap + count) of gen[d & 1].  Every kernel that
// continues a path (shade, scatter) writes the whole record at the path's position in the NEXT generation, so a bounce reads and
// writes contiguous memory however few of the pass's paths are still alive (round 1 indexed these arrays by the fixed path slot:
// deep bounces then used a fraction of every 128-byte line — 1.8x the touched bytes in k_shade, 2.8x in k_shadow).
// 1: a path's wavelengths live once, by camera slot (DPathState::lambda_s); 0: every generation record carries a copy (DPathGen::lambda)
#ifndef HK_LAMBDA_BY_SLOT
#define HK_LAMBDA_BY_SLOT 1
#endif
struct DPathGen {
    float4* ray_o;         // o.xyz, t_max
    float4* ray_d;         // d.xyz, time
    float4* beta;
    float4* r_u;
    float4* r_l;
    float4* lambda;
    uint2* meta;           // x = flags: depth(8) | specular(1)<<8 | any_non_specular(1)<<9 | (medium+1)<<16;  y = path slot   (DPathState::meta32: one uint32 per record, see there)
};

struct DPathState {
    int capacity;          // path slots (pixel x sample-in-pass)
    int n_waves;           // W: every queue is split into W wave-private segments
    int wave_cap;          // entries per wave segment (multiple of 64); W * wave_cap >= capacity
    DPathGen gen[2];       // gen[depth & 1]: the ray queue of that depth IS this array (no index queue)
    float4* hit;           // per entry of the current generation: t, prim(bits), u, v
    int* mat_id;           // per entry of the current generation: resolved material index of the hit (| HK_MAT_EMISSIVE_BIT)
    uint2* sel_light;      // per entry of the current generation: the next-event light k_light_select chose (1-based index, pmf bits); scenes without media
    // per path slot: written once by the camera kernel, accumulated into (L) along the path, read by the film kernel
    float4* lambda_s;
    float4* pdf;           // unused since round 3: the wavelength pdfs are recomputed from lambda_s by k_film (kept: ABI of the struct)
    float4* L;
    float* filter_w;
    // shadow records, dense in shadow-queue order (entry = segment * wave_cap + position in the segment's shadow queue)
    float4* sh_o;          // shadow ray: o.xyz, t_max
    float4* sh_d;          // d.xyz, medium (bits)
    float4* sh_Ld;
    float4* sh_ru;
    float4* sh_rl;
    uint32_t* sh_slot;     // path slot whose L receives the contribution
    // split shadow walk of grey media (k_walk_cast / k_walk_track; null when the scene has no medium): a ray's state between the two
    float4* sh_T;          // T_ray, r_u, r_l (one value for all four wavelengths), hit_t of the pending medium segment
    uint32_t* sh_aux;      // segments | miss << 8 | transition << 9 | (next medium + 1) << 16
    float4* sh_it;         // 4 per record: the tracker's majorant iterator (next_t | t_min, delta_t | t_max, voxel | mode) and PCG32 state, set up by the cast
    uint32_t* wq_a;        // global index queue: records that need a cast (rounds > 0)
    uint32_t* wq_b;        // global index queue: records that need a tracking pass
    int* wq_ctl;           // [depth][round][4]: count A, cursor A, count B, cursor B
    // index queues: entries are generation indices p of the current depth
    uint32_t* escaped_q;
    uint32_t* medium_q;    // rays that travel inside a medium (delta tracking before their surface hit is processed)
    uint32_t* scatter_q;   // paths that scattered inside a medium at this depth (K5/K6 input)
    int* initial_medium;   // camera medium detected on the device (K14)
    uint32_t* mat_q;       // HK_MAX_KINDS * W * wave_cap
    int* counters;         // [(max_depth + 2) * Q_COUNT][W] per-wave queue sizes
    int* tickets;          // [ticket_rows * HK_TICKET_COLS][HK_TICKET_WAYS * HK_TICKET_STRIDE] segment tickets (dynamic segment -> wave assignment), zeroed per pass
    int ticket_rows;       // max_depth + 2
    int* seg_list;         // [depth][queue][n_waves]: ascending non-empty segments of each queue (k_segment_lists), laid out like `counters`
    int* seg_list_n;       // [depth][queue]: their number
    int dynamic_segments;  // 1: every kernel draws its segments from the tickets (scenes with media); 0: static stride
    int small_pass;        // 1: a pass of at most one segment per resident wave (a one-sample call): wave g takes segment g — no work lists, no tickets
    int ticket_share;      // 1: waves publish exhausted ticket counters in one shared word per ticket and read it when they open one (HK_TICKET_SHARE=0: off)
    // 1 (scenes without media): r_u is 1 on every path and the four components of r_l are equal (r_l = r_u / pdf of a scalar pdf,
    // only media rescale them per wavelength) — r_u is not stored, r_l is one float per record (the float4 array's memory, read as
    // float), and a shadow record carries its two MIS weights as one float2 in sh_ru.  Same arithmetic on the broadcast values.
    int compact;
    // LEAN GENERATION RECORDS (round 6; opaque scenes without media, HK_LEAN_RECORDS=0: off).  meta32: a record's meta word is ONE uint32 — the path
    // slot (< 2^30) with the specular / any-non-specular flags in its two top bits; the depth is the kernel's own `depth` and there is no
    // medium (4 B instead of 8 per record written and read).  const_origin (pinhole camera): every camera ray starts at the same point, so
    // generation 0 stores ONE ray_o per segment (its first entry) that every depth-0 reader of the segment loads (16 B less per path written by
    // k_camera and fetched by the depth-0 traversal and shading).  Independently of both, the COMPACT layouts keep r_l — one float — in the
    // unused fourth word of ray_d instead of an array of its own.
    int meta32, const_origin;
    int sh_final;          // 1 (opaque scenes without media): slim shadow records — sh_d.w holds the denominator average(w_u + w_l) of the contribution, the two weights are not stored (hk_kernels.hip: shadow_contribute_final)
};

struct DStats {
    unsigned long long rays_closest, rays_shadow, nodes, tris, hits, vertices, collisions, light_nodes, sh_nodes, sh_tris;
    unsigned long long nvdb_collisions, sh_nvdb_collisions;   // the part of collisions / sh_collisions counted while a scene with a NanoVDB medium was rendered
    unsigned long long sh_collisions, dda_steps, sh_dda_steps, scatter_vertices, sc_light_nodes;   // ratio-tracking collisions of the shadow walk; majorant cells entered (delta / ratio tracking); K5+K6 vertices
#ifdef HK_DEBUG_UTIL   // lane-utilisation bookkeeping of the media state machines (debug builds only: tools/build_variant.sh dbg -DHK_DEBUG_UTIL)
    unsigned long long dbg[32];
#endif
};

struct DFrame {            // per-pass constants
    int width, height, tiles_x, tiles_y;   // film size; 8x8 pixel tiles of the rendered pixel range
    int x0, y0, x1, y1;    // rendered pixel range [x0, x1) x [y0, y1) (0-based; the whole film unless hk_render_tile narrows it)
    int n_pixels_padded;   // tiles_x*tiles_y*64
    int samples_in_pass;
    uint32_t s_mul;        // slot / samples_in_pass == (slot * s_mul) >> s_shr for slot < 2^30 (path slot = pixel slot * samples_in_pass + k)
    int s_shr;
    int first_sample, sample_stride;   // sample index of pass-sample k = first_sample + k*sample_stride
    int max_depth;
    int regularize;
    float max_component_value;
    int count_nodes;       // 1: accumulate node/triangle counters
    int implicit_ones;     // 1: scene without media: the depth-0 records do not store beta = r_u = r_l = 1
    int track_gate;        // k_track_flat: collision rounds wait until this many lanes hold a tentative collision | cap on the extra cheap steps << 8 (HK_TRACK_MIN_PENDING, HK_TRACK_EXTRA_ADVANCE)
    int delta_advance;     // k_track: cheap steps (next majorant cell / free-flight sample) per round before the pending collisions are evaluated (HK_DELTA_ADVANCE)
    int refill_idle;       // k_track: idle lanes that trigger a refill round (HK_TRACK_REFILL_IDLE)
    int walk_tune;         // k_shadow_walk: tracking batches per round | advance steps per batch << 8 | feed rounds << 16 | idle lanes that trigger a refill << 24 (HK_SHADOW_TRACK_BATCH, HK_TRACK_ADVANCE, HK_SHADOW_FEED_ROUNDS, HK_WALK_REFILL_IDLE)
};

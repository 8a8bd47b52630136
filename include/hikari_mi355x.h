/*
 * hikari_mi355x.h — C ABI of the MI355X-native VolPath hot path.
 *
 * This is the drop-in boundary for ONE path of JuliaGraphics/Hikari.jl: the per-ray inner loop
 * of the `VolPath` integrator.  The reference has no FFI for it: callers invoke
 *
 *     integrator(scene, film, camera)          src/integrators/volpath/volpath.jl:655-670
 *     render!(integrator, scene, film, camera) src/integrators/volpath/volpath.jl:445-636
 *     clear!(integrator)                       src/integrators/volpath/volpath.jl:108-113
 *     close(integrator)                        src/Hikari.jl:47
 *
 * A Julia shim (julia/HikariMI355X.jl) subtypes `Hikari.Integrator`, flattens
 * `Scene`/`Film`/`Camera` into the POD records below and `ccall`s these entry points.
 *
 * Conventions
 *  - every entry point returns int32 status: 0 = ok, <0 = error; hk_last_error() gives the text.
 *  - no exceptions cross the boundary; one hk_ctx is used from one thread at a time.
 *  - host input buffers are borrowed for the duration of the call and copied to the device.
 *  - indices inside records are 0-based unless the field name says `_1based`; -1 means "none".
 *  - arrays described as "Julia layout" are column-major exactly as the Julia arrays they replace.
 *  - all floating point is IEEE binary32 unless stated.
 */
#ifndef HIKARI_MI355X_H
#define HIKARI_MI355X_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HK_OK 0
#define HK_ERR_INVALID -1
#define HK_ERR_DEVICE -2
#define HK_ERR_UNSUPPORTED -3
#define HK_UNSET -100 /* hk_ctx_get_option: the knob has no value (distinct from every error code) */

/* ---------------------------------------------------------------------------------------------
 * Textures (reference: textures/texture-ref.jl:151-186 bilinear fetch with the (1-v, u) flip).
 * data = Julia Matrix [height, width] of `channels` floats per texel (column-major: texel (iy,ix)
 * at ((iy + height*ix) * channels)).  channels is 1 (Float32 texture) or 4 (RGBSpectrum r,g,b,a).
 * ------------------------------------------------------------------------------------------- */
typedef struct hk_texture {
    int32_t width, height, channels;
    int32_t kind; /* 0: 2-D image (above).  1: VertexColorTexture (textures/basic.jl:42-46; texture-ref.jl:230-235):
                     data = face_colors[3, n_faces] (Julia layout: the 3 colours of a face are adjacent), height = 3,
                     width = n_faces; looked up with the hit's face index and barycentrics, not with uv */
    const float* data;
} hk_texture;

/* A material parameter that is either a constant or a texture reference (texture-ref.jl:50-84). */
typedef struct hk_tex_rgba {
    float c[4];  /* constant r,g,b,alpha (RGBSpectrum is 4 floats, spectrum.jl:38-43) */
    int32_t tex; /* -1 => constant `c`; otherwise index into hk_scene_desc.textures */
} hk_tex_rgba;

typedef struct hk_tex_f32 {
    float v;
    int32_t tex;
} hk_tex_f32;

/* ---------------------------------------------------------------------------------------------
 * Materials (Appendix A of SURVEY.md; uber-material.jl:180-215,378-384; coated-*.jl;
 * thin-dielectric.jl:45-47; diffuse-transmission.jl:39-43; mix-material.jl).
 * One tagged record; which slots a kind reads is listed next to the kind.
 * ------------------------------------------------------------------------------------------- */
enum {
    HK_MAT_MATTE = 0,            /* rgb[0]=Kd, f[0]=sigma */
    HK_MAT_MIRROR = 1,           /* rgb[0]=Kr */
    HK_MAT_GLASS = 2,            /* rgb[0]=Kr, rgb[1]=Kt, f[0]=index (roughness ignored, Q12) */
    HK_MAT_CONDUCTOR = 3,        /* rgb[0]=eta, rgb[1]=k (or spectrum[0/1]), f[0]=roughness, flags&REMAP */
    HK_MAT_COATED_DIFFUSE = 4,   /* rgb[0]=reflectance, rgb[1]=albedo, f[0]=u_rough, f[1]=v_rough,
                                    f[2]=thickness, f[3]=eta, f[4]=g, i[0]=max_depth, i[1]=n_samples */
    HK_MAT_THIN_DIELECTRIC = 5,  /* f[0]=eta */
    HK_MAT_DIFFUSE_TRANSMISSION = 6, /* rgb[0]=reflectance, rgb[1]=transmittance, f[0]=scale */
    HK_MAT_COATED_DIFFUSE_TRANSMISSION = 7, /* rgb[0]=reflectance, rgb[1]=transmittance, rgb[2]=albedo,
                                    f[0..4] as coated diffuse, i[0], i[1] */
    HK_MAT_COATED_CONDUCTOR = 8, /* rgb[0]=conductor eta, rgb[1]=conductor k, rgb[2]=reflectance,
                                    rgb[3]=albedo, f[0]=iface u_rough, f[1]=iface v_rough, f[2]=iface eta,
                                    f[3]=cond u_rough, f[4]=cond v_rough, f[5]=thickness, f[6]=g,
                                    i[0]=max_depth, i[1]=n_samples, flags&USE_ETA_K */
    HK_MAT_MIX = 9,              /* f[0]=amount, i[0]=material1, i[1]=material2 (indices into materials), mix_key */
    HK_MAT_FALLBACK = 10         /* any other Material (e.g. a bare Emissive): gray 0.5 Lambertian, Q24 */
};
#define HK_MATF_REMAP_ROUGHNESS 1
#define HK_MATF_USE_ETA_K 2

typedef struct hk_material {
    int32_t kind;
    int32_t flags;
    hk_tex_rgba rgb[4];
    hk_tex_f32 f[8];
    int32_t i[4];
    int32_t spectrum[2];  /* piecewise-linear spectrum ids for eta / k, -1 => use rgb slot */
    /* Mix only: the SetKeys of the two sub-materials as the reference hashes them
       (mix-material.jl:96-127): {type_idx1, vec_idx1, type_idx2, vec_idx2}. */
    uint32_t mix_key[4];
} hk_material;

/* PiecewiseLinearSpectrum (spectral/piecewise-linear.jl; metal-spectra.jl): n (lambda, value) knots. */
typedef struct hk_pl_spectrum {
    int32_t n;
    int32_t _pad;
    const float* lambdas;
    const float* values;
} hk_pl_spectrum;

/* MediumInterfaceIdx (materials/medium-interface.jl:76-80). */
typedef struct hk_medium_interface {
    int32_t material; /* index into materials */
    int32_t inside;   /* index into media, -1 = vacuum */
    int32_t outside;
} hk_medium_interface;

/* TriangleMeta (scene.jl:11-15). */
typedef struct hk_tri_meta {
    uint32_t medium_interface_idx; /* 0-based index into media_interfaces */
    uint32_t primitive_index;      /* face index within its mesh, as the reference stores it (1-based) */
    uint32_t arealight_flat_idx_1based; /* flat index into lights, 0 = no area light */
} hk_tri_meta;

/* ---------------------------------------------------------------------------------------------
 * Lights (src/lights; flat order = the reference's flat light index, light-sampler.jl:289-329).
 * ------------------------------------------------------------------------------------------- */
enum {
    HK_LIGHT_POINT = 0,
    HK_LIGHT_SPOT = 1,
    HK_LIGHT_DIRECTIONAL = 2,
    HK_LIGHT_SUN = 3,
    HK_LIGHT_AMBIENT = 4,
    HK_LIGHT_ENVIRONMENT = 5,
    HK_LIGHT_DIFFUSE_AREA = 6
};
enum {
    HK_SPEC_RGB = 0,       /* RGBSpectrum: uplifted at run time with uplift_rgb_illuminant */
    HK_SPEC_ILLUMINANT = 1 /* RGBIlluminantSpectrum{poly(c0,c1,c2), scale} (rgb2spec.jl:317-335) */
};

typedef struct hk_light {
    int32_t kind;
    int32_t spectrum_kind;
    float i_rgb[4];       /* RGBSpectrum i / scale (environment) */
    float poly[3];        /* RGBIlluminantSpectrum coefficients */
    float illum_scale;    /* RGBIlluminantSpectrum.scale */
    float scale;          /* light.scale (Q4) */
    float position[3];    /* point, spot */
    float direction[3];   /* directional, sun: travel direction, normalised */
    float world_to_light[16]; /* spot, row-major 4x4 */
    float light_to_world[16];
    float cos_total_width, cos_falloff_start; /* spot */
    /* diffuse area (lights/diffuse-area.jl:25-33) */
    float v[9];
    float normal[3];
    float area;
    float uv[6];
    hk_tex_rgba Le;
    int32_t two_sided;
    int32_t envmap; /* environment: index into envmaps */
} hk_light;

/* EnvironmentMap + Distribution2D (textures/environment_map.jl:9-45; sampler/sampling.jl:179-262). */
typedef struct hk_envmap {
    int32_t width, height;    /* data is Julia Matrix{RGBSpectrum}[height,width], 4 floats per texel */
    const float* data;
    float rotation[9];        /* 3x3 row-major */
    int32_t nu, nv;
    const float* conditional_func;     /* [nu, nv]   Julia layout */
    const float* conditional_cdf;      /* [nu+1, nv] */
    const float* conditional_func_int; /* [nv] */
    const float* marginal_func;        /* [nv] */
    const float* marginal_cdf;         /* [nv+1] */
    float marginal_func_int;
    int32_t _pad;
} hk_envmap;

/* ---------------------------------------------------------------------------------------------
 * Media (volpath/media.jl, nanovdb.jl) — records are defined now so the ABI is stable.
 * ------------------------------------------------------------------------------------------- */
enum { HK_MEDIUM_HOMOGENEOUS = 0, HK_MEDIUM_GRID = 1, HK_MEDIUM_RGB_GRID = 2, HK_MEDIUM_NANOVDB = 3 };

typedef struct hk_medium {
    int32_t kind;
    float sigma_a[4], sigma_s[4], Le[4]; /* RGBSpectrum */
    float g;
    float sigma_scale, Le_scale;
    /* grid media */
    float bounds_min[3], bounds_max[3];
    float render_to_medium[16], medium_to_render[16]; /* row-major */
    int32_t res[3];
    const float* density;      /* Grid: [nx,ny,nz] Julia layout (x fastest) */
    const float* sigma_a_grid; /* RGBGrid: rgba [nx,ny,nz] or NULL */
    const float* sigma_s_grid;
    const float* Le_grid;
    int32_t majorant_res[3];
    const float* majorant;     /* [rx*ry*rz], index x + rx*(y + ry*z) */
    float max_density;
    /* NanoVDB */
    const uint8_t* nvdb_bytes;
    int64_t nvdb_size;
    int64_t root_offset_1based, upper_offset_1based, lower_offset_1based, leaf_offset_1based;
    int32_t upper_count, lower_count, leaf_count, root_table_size;
    float inv_mat[9], vec[3];
    int32_t index_bbox_min[3], index_bbox_max[3];
} hk_medium;

/* ---------------------------------------------------------------------------------------------
 * Scene: flat triangle soup (world space) + records.  Triangle t has vertices
 * positions[9t..9t+8]; normals/uvs/tangents follow the same [T][3][k] layout, NULL = absent
 * (NaN normals/tangents per triangle also mean absent: intersection.jl:136-139,93).
 * Absent uvs default to (0,0),(1,0),(1,1).
 * ------------------------------------------------------------------------------------------- */
typedef struct hk_scene_desc {
    int32_t n_triangles;
    int32_t n_materials;
    int32_t n_textures;
    int32_t n_media_interfaces;
    int32_t n_lights;
    int32_t n_envmaps;
    int32_t n_media;
    int32_t n_spectra;
    const float* positions;
    const float* normals;
    const float* uvs;
    const float* tangents;
    const hk_tri_meta* meta;
    const hk_material* materials;
    const hk_texture* textures;
    const hk_medium_interface* media_interfaces;
    const hk_light* lights;
    const hk_envmap* envmaps;
    const hk_medium* media;
    const hk_pl_spectrum* spectra;
} hk_scene_desc;

/* Data tables the reference embeds in its sources; the caller hands them over once per context.
 * sobol: SobolMatrices32 (sampler/sobol_matrices.jl, 1024 x 52 uint32; only dims 0,1 are read)
 * cie:   CIE_X, CIE_Y, CIE_Z 1-nm tables 360..830 (spectral/color.jl:53-345)
 * rgb2spec: RGBToSpectrumTable (spectral/rgb2spec.jl:71-75): coeffs[3,res,res,res,3] Julia layout. */
typedef struct hk_tables {
    const uint32_t* sobol_matrices;
    int32_t sobol_count; /* number of uint32 (>= 104) */
    int32_t rgb2spec_res;
    const float* cie_x;
    const float* cie_y;
    const float* cie_z; /* 471 floats each */
    const float* rgb2spec_scale;
    const float* rgb2spec_coeffs;
} hk_tables;

/* VolPath parameters (volpath.jl:29-101) + pixel filter (filter.jl:574-604). */
enum { HK_FILTER_BOX = 1, HK_FILTER_TRIANGLE = 2, HK_FILTER_GAUSSIAN = 3, HK_FILTER_MITCHELL = 4, HK_FILTER_LANCZOS = 5 };
enum { HK_COHERENCE_NONE = 0, HK_COHERENCE_SORTED = 1, HK_COHERENCE_PER_TYPE = 2 };

typedef struct hk_integrator_params {
    int32_t max_depth;              /* default 8 */
    int32_t samples_per_pixel;      /* default 64 */
    int32_t russian_roulette_depth; /* carried, unused by the kernels (Q7) */
    int32_t regularize;             /* default 1 */
    int32_t material_coherence;     /* result-neutral knob; the HIP path always sorts by material kind */
    float max_component_value;      /* default 10 */
    int32_t filter_type;            /* default HK_FILTER_GAUSSIAN */
    float filter_radius[2];         /* default 1.5, 1.5 */
    float filter_param1;            /* sigma (Gaussian, default 0.5) | B (Mitchell) | tau (Lanczos) */
    float filter_param2;            /* C (Mitchell) */
    int32_t accumulate_f64;         /* accumulation_eltype == Float64 */
    uint32_t sampler_seed;          /* reference fixes 0 (volpath.jl:480) */
    int32_t samples_per_pass;       /* HIP-only tuning: samples in flight per wavefront pass (0 = auto) */
} hk_integrator_params;

/* One record covers PerspectiveCamera (camera/perspective.jl:41-128) and MatrixCamera
 * (camera/matrix.jl:13-58, lens_radius = 0).  Matrices are row-major 4x4. */
typedef struct hk_camera {
    float raster_to_camera[16];
    float camera_to_world[16];
    float lens_radius, focal_distance;
    float shutter_open, shutter_close;
    float dx_camera[3], dy_camera[3];
} hk_camera;

typedef struct hk_stats {
    uint64_t rays_closest;     /* closest-hit casts from camera/indirect segments (K3) */
    uint64_t rays_shadow;      /* closest-hit casts from shadow segments (K10) */
    uint64_t bvh_nodes_visited;
    uint64_t tris_tested;
    uint64_t hits_accepted;
    uint64_t path_vertices;    /* surface shading events */
    uint64_t medium_collisions;
    uint64_t light_bvh_nodes;
    double seconds_trace;      /* HIP-event time spent in the closest-hit traversal kernel (k_trace) */
    double seconds_total;      /* HIP-event span from the first hk_render after hk_stats_reset to the last */
    uint64_t trace_launches;
    /* per-kernel-class breakdown (filled when hk_stats_enable_counters flags are set) */
    uint64_t trace_nodes, trace_tris;    /* k_trace only (needs flag bit 0) */
    uint64_t shadow_nodes, shadow_tris;  /* k_shadow only */
    uint64_t shadow_launches, shade_launches;
    double seconds_shadow, seconds_shade, seconds_other; /* HIP-event sums (flag bit 1) */
    /* ALGORITHMIC bytes of the three hot kernel classes, SURVEY 8(d): per cast 32 (ray in) + 64 per BVH node visited + 36 per
       triangle tested + 96 per accepted hit (closest-hit only) + 16 (hit out); per path vertex 2*104 (state read + written) + 64
       (material record) + 96 (light record) + 60 per light-BVH node.  The node / triangle terms need counter flag bit 0. */
    uint64_t bytes_algorithmic_trace, bytes_algorithmic_shadow, bytes_algorithmic_shade;
    double seconds_media;      /* HIP-event sum of the delta-tracking kernels (k_track + k_scatter); not part of seconds_other */
    /* media class (84 B per collision through a NanoVDB tree / 36 B on a dense grid, 4 B per majorant cell entered).
       medium_collisions above = track_collisions + shadow_collisions. */
    uint64_t track_collisions, shadow_collisions;   /* tentative collisions of delta tracking (K4) / of the shadow rays' ratio tracking (K10) */
    uint64_t track_dda_steps, shadow_dda_steps;     /* majorant cells entered by K4 / by K10 */
    uint64_t scatter_vertices;                      /* real scattering events handed to K5 + K6 */
    uint64_t media_launches;                        /* k_track + k_scatter launches */
    /* collisions x 84|36 + 4 x DDA steps of K4, + (2 x 104 state + 96 light record) per scattering vertex of K5 + K6;
       bytes_algorithmic_shadow likewise includes the shadow walk's collisions and DDA steps */
    uint64_t bytes_algorithmic_media;
    /* the part of seconds_shade spent choosing the next-event light in a kernel of its own (scenes with a deep light BVH: k_light_select*),
       and its launches (not part of shade_launches) */
    double seconds_select;
    uint64_t select_launches;
    /* passes rendered as ONE launch (k_small_pass: camera rays and every bounce of a small pass of a closed all-matte scene; the film
       kernel follows).  Their stages are not counted in *_launches. */
    uint64_t fused_passes;
} hk_stats;

typedef struct hk_ctx hk_ctx;
typedef struct hk_scene hk_scene;
typedef struct hk_film hk_film;
typedef struct hk_integrator hk_integrator;

/* Context: one per GPU.  `stream` is a hipStream_t (0 = default stream); it lets the host harness
 * run the path on a stream it owns (torch.cuda.Stream().cuda_stream). */
int32_t hk_ctx_create(int32_t device_id, void* stream, hk_ctx** out);
int32_t hk_ctx_destroy(hk_ctx* ctx);
const char* hk_last_error(void);
int32_t hk_ctx_set_tables(hk_ctx* ctx, const hk_tables* tables);
/* Tuning knobs (the HK_* table of INTEGRATION.md).  The library reads the environment ONCE, in hk_ctx_create — never on a render
 * path — and keeps the values in the context; hk_ctx_set_option changes one afterwards (value NULL: back to the built-in default).
 * Calls that were only noted (see hk_render) are rendered first, under the options they were made with.  Unknown names are
 * HK_ERR_INVALID.  hk_ctx_get_option: length of the value (copied to `out`, NUL-terminated, at most out_bytes); HK_UNSET — not an error:
 * the built-in default applies, `out` becomes the empty string — when the knob has no value.
 * There is no reference counterpart: Hikari's knobs are keyword arguments of VolPath(...) (volpath.jl:55-106), carried here by
 * hk_integrator_params. */
int32_t hk_ctx_set_option(hk_ctx* ctx, const char* name, const char* value);
int32_t hk_ctx_get_option(hk_ctx* ctx, const char* name, char* out, int32_t out_bytes);
/* Gives the path-state slabs this process keeps cached for `ctx`'s device back to the driver (a destroyed integrator's path
 * state — up to HK_STATE_CACHE_GB = 128 in total — is otherwise kept for the next one: allocating 84 GB takes seconds). */
int32_t hk_trim_cache(hk_ctx* ctx);

/* Scene: copies the description, builds the BVH + light BVH on the host, uploads. */
int32_t hk_scene_create(hk_ctx* ctx, const hk_scene_desc* desc, hk_scene** out);
int32_t hk_scene_destroy(hk_scene* scene);

int32_t hk_integrator_create(hk_ctx* ctx, const hk_integrator_params* params, hk_integrator** out);
int32_t hk_integrator_destroy(hk_integrator* integ);

/* Film accumulators [pixel_rgb 3N | pixel_weight_sum N] (volpath-state.jl:122-131).  If
 * `external_accum` is non-NULL it is a device pointer to 4*N floats (or doubles when the integrator
 * accumulates in f64) owned by the caller — e.g. a torch tensor that torch.distributed reduces. */
int32_t hk_film_create(hk_ctx* ctx, int32_t width, int32_t height, int32_t f64, void* external_accum, hk_film** out);
int32_t hk_film_destroy(hk_film* film);
int32_t hk_film_clear(hk_film* film); /* clear!(vp), volpath.jl:108-113 */

/* Render samples first_sample_idx .. first_sample_idx+n_samples-1 (1-based like
 * film.iteration_index, volpath.jl:488-489) with stride `sample_stride` (1 on a single GPU; G when
 * G GPUs shard by sample index) on top of the current accumulators.
 *
 * ORDERING CONTRACT.  The call returns before the samples are rendered.
 *   - A call of more than 8 M paths (HK_PIPELINE_MAX_PATHS_M), a call on a context that was created on a CALLER'S stream, and a call
 *     into a film with EXTERNAL accumulators — or one whose accumulators hk_film_accum_device_ptr has handed out — are enqueued on ctx's stream before the call returns: whatever the caller orders behind
 *     that stream (events, a collective on the accumulators, torch.cuda.synchronize()) sees the samples.
 *   - A SMALL call (a one-sample `render!`, volpath.jl:445-450) on a context with the default stream and a library-owned film may only
 *     be NOTED: calls that continue each other (same scene / integrator / film / camera / pixel range / stride, sample indices
 *     following on) are rendered as ONE pass — bit-identical film, a seventh of the time — when the note reaches HK_BATCH_PATHS_M
 *     (64 M paths), when a call comes that does not continue it, or when anything looks: hk_flush, hk_sync, every hk_film_* / hk_stats_* /
 *     hk_*_destroy / hk_ctx_set_option entry point.  hk_flush enqueues the noted calls without waiting for them.
 *     HK_BATCH_PATHS_M=0 (hk_ctx_set_option) turns the noting off.
 * Argument errors are reported by the call itself; a device error of a deferred pass by the call that flushes it — also by the
 * destroy entry points, which still destroy their object. */
int32_t hk_render(hk_ctx* ctx, hk_scene* scene, hk_integrator* integ, hk_film* film, const hk_camera* cam,
                  int32_t first_sample_idx, int32_t n_samples, int32_t sample_stride);

/* The same, restricted to the pixels [x0, x1) x [y0, y1) (0-based, py counted like the film rows of hk_film_read_rgb): the
 * pixel-tile sharding mode of SURVEY 8(e).  Paths are independent (volpath.jl:538-612 reads no other pixel) and ZSobol is a
 * pure function of (px, py, sample_idx, dim), so a pixel gets bit-identical accumulators whichever range it is rendered in;
 * pixels outside the range are left untouched, so the sum-reduce of zero-initialised films gathers disjoint tiles. */
int32_t hk_render_tile(hk_ctx* ctx, hk_scene* scene, hk_integrator* integ, hk_film* film, const hk_camera* cam,
                       int32_t first_sample_idx, int32_t n_samples, int32_t sample_stride, int32_t x0, int32_t y0, int32_t x1, int32_t y1);

/* ---------------------------------------------------------------------------------------------
 * Multi-GPU: units = (pixel, sample_idx) paths, no communication between paths; each device renders its share
 * (sample indices g+1, g+1+G, ... through `sample_stride`, or a pixel tile through hk_render_tile) of the same replicated scene,
 * and ONE exchange finishes the frame: ncclReduce(sum) of [pixel_rgb 3N | pixel_weight_sum N] (volpath.jl:364-373,
 * volpath-state.jl:122-131) onto the root over xGMI.  RCCL is loaded on first use.
 *   one process, several GPUs (the Julia shim's `devices = 0:7`): hk_ctx per device, hk_comm_create over them, hk_render on
 *     each, hk_film_reduce(comm, films, n, root), hk_film_read_rgb on films[root];
 *   one process per GPU (torchrun-style): rank 0 calls hk_comm_unique_id and hands the 128 bytes to the others (any side
 *     channel), every rank hk_comm_create_rank, then hk_film_reduce(comm, &film, 1, root).
 * The reduce is enqueued on each context's stream, after the renders already enqueued there, in place.
 * ------------------------------------------------------------------------------------------- */
typedef struct hk_comm hk_comm;
int32_t hk_comm_create(hk_ctx* const* ctxs, int32_t n, hk_comm** out);
int32_t hk_comm_unique_id(uint8_t* id_out128);
int32_t hk_comm_create_rank(hk_ctx* ctx, const uint8_t* id128, int32_t rank, int32_t world, hk_comm** out);
int32_t hk_comm_destroy(hk_comm* comm);
int32_t hk_film_reduce(hk_comm* comm, hk_film* const* films, int32_t n_films, int32_t root);

/* K13 finalize (volpath.jl:384-417): writes rgb/weight as Julia Matrix{RGB{Float32}}[height,width]
 * (`out_hw3` = 3 floats per pixel, column-major over (py,px)) into host memory; synchronises. */
int32_t hk_film_read_rgb(hk_ctx* ctx, hk_film* film, float* out_hw3);
/* The same frame without stopping the GPU for the copy — what an interactive viewer wants between two render! calls
 * (volpath.jl:617-633 writes the frame after EVERY sample):
 *   hk_film_read_rgb_async   enqueues K13 + the copy of the frame into PINNED host memory owned by the film (two buffers, used in
 *                            turn) behind everything rendered so far, and returns at once — the next hk_render may follow;
 *   hk_film_read_wait        waits for the LAST hk_film_read_rgb_async of the film only (not for renders enqueued after it), then
 *                            copies the frame to out_hw3 (may be NULL) and / or returns the pinned buffer itself in *frame (may be
 *                            NULL; valid until the second-next hk_film_read_rgb_async of this film).
 * hk_film_read_rgb itself goes through the same pinned buffers and a memcpy.  A viewer that reads every frame into ONE buffer can
 * spare the memcpy: hk_film_pin_host registers that buffer (3 * width * height floats) with the driver (hipHostRegister) and
 * hk_film_read_rgb copies straight into it whenever it is the destination; the caller keeps the buffer alive until hk_film_unpin_host
 * or hk_film_destroy.  The library never registers caller memory on its own (it cannot know when such a buffer is freed), unless
 * HK_READBACK_PIN=1 asks for round 5's behaviour: the same destination twice in a row is registered. */
int32_t hk_film_read_rgb_async(hk_ctx* ctx, hk_film* film);
int32_t hk_film_read_wait(hk_ctx* ctx, hk_film* film, float* out_hw3, const float** frame);
int32_t hk_film_pin_host(hk_film* film, float* host_hw3);
int32_t hk_film_unpin_host(hk_film* film);
/* raw accumulators (host copy): 4*N floats (or doubles). */
int32_t hk_film_read_accum(hk_ctx* ctx, hk_film* film, void* out);
/* the accumulators on the device.  Noted calls are enqueued first, and because the caller may KEEP the pointer (a reduce, an event of
 * its own behind ctx's stream), every later hk_render into this film is enqueued before it returns — as for external accumulators. */
void* hk_film_accum_device_ptr(hk_film* film);

/* hk_flush: every render call made so far is enqueued on ctx's stream (noted small calls are rendered now); does not wait.
 * hk_sync: hk_flush + waits until the stream is idle. */
int32_t hk_flush(hk_ctx* ctx);
int32_t hk_sync(hk_ctx* ctx);
int32_t hk_stats_get(hk_ctx* ctx, hk_stats* out);
int32_t hk_stats_reset(hk_ctx* ctx);
/* flags bit 0: count BVH nodes/triangles per cast during the next renders (adds two counters per lane; keep
 * off while timing); bit 1: bracket every kernel launch with HIP events on ctx's stream. */
int32_t hk_stats_enable_counters(hk_ctx* ctx, int32_t flags);

/* ---- sub-kernel entry points (parity tests drive these through the same ABI) ---- */
/* closest-hit of n rays (host arrays): t (inf = miss), prim (index into scene triangles, -1 = miss), bary u,v */
int32_t hk_trace_closest(hk_ctx* ctx, hk_scene* scene, int32_t n, const float* o3, const float* d3, const float* tmax,
                         float* out_t, int32_t* out_prim, float* out_uv2);
/* ZSobol draws on the device: for i<n: out[i] = sample_1d / sample_2d(px[i],py[i],sample_idx[i],dim[i]) */
int32_t hk_test_sobol(hk_ctx* ctx, int32_t width, int32_t height, int32_t spp, uint32_t seed, int32_t n,
                      const int32_t* px, const int32_t* py, const int32_t* sample_idx, const int32_t* dim,
                      float* out_1d, float* out_2d);
/* camera-sample stage (K1) for n (px,py,sample): lambda[4],pdf[4],filter weight, ray o,d  => 15 floats each */
int32_t hk_test_camera(hk_ctx* ctx, hk_integrator* integ, const hk_camera* cam, int32_t width, int32_t height, int32_t n,
                       const int32_t* px, const int32_t* py, const int32_t* sample_idx, float* out15);
/* uplift: mode 0 bounded, 1 unbounded, 2 illuminant; rgb[3n], lambda[4n] -> out[4n] */
int32_t hk_test_uplift(hk_ctx* ctx, int32_t mode, int32_t n, const float* rgb, const float* lambda, float* out);
/* light-BVH: sample (light_idx_1based, pmf) and pmf of a given light for n shading points */
int32_t hk_test_light_bvh(hk_ctx* ctx, hk_scene* scene, int32_t n, const float* p3, const float* n3, const float* u,
                          int32_t* out_light, float* out_pmf, const int32_t* query_light, float* out_query_pmf);
/* ---------------------------------------------------------------------------------------------
 * postprocess!(film; exposure, tonemap, gamma, white_point, sensor, background)  (src/postprocess.jl:293-357).
 * Non-destructive: reads the film's current rgb/weight, writes tonemapped RGB to `dst` (host, Julia [h,w] RGB{Float32}
 * layout like hk_film_read_rgb).  Parameters are the kernel arguments of postprocess_kernel! (:185-250): the host side
 * resolves symbols / sensor / white-balance matrix (compute_white_balance_matrix, spectral/color.jl:522-547).
 * depth: optional host float[h*w] (film.depth, Julia [h,w]); with mask_escaped != 0 pixels are blended towards `bg` by the
 * fraction of +Inf depths in their 3x3 neighbourhood.
 * ------------------------------------------------------------------------------------------- */
enum { HK_TONEMAP_NONE = 0, HK_TONEMAP_REINHARD = 1, HK_TONEMAP_REINHARD_EXT = 2, HK_TONEMAP_ACES = 3, HK_TONEMAP_UNCHARTED2 = 4, HK_TONEMAP_FILMIC = 5 };
typedef struct hk_postprocess_params {
    float exposure;
    int32_t tonemap;
    float inv_gamma;
    int32_t apply_gamma;
    float white_point;
    float imaging_ratio; /* sensor.exposure_time * sensor.iso / 100 */
    int32_t apply_wb;
    float wb[9];         /* row-major 3x3 Bradford matrix */
    int32_t mask_escaped;
    float bg[3];
} hk_postprocess_params;
int32_t hk_film_postprocess(hk_ctx* ctx, hk_film* film, const hk_postprocess_params* params, const float* depth, float* dst_rgb);
/* the same kernel on a caller-supplied framebuffer (host, Julia [h,w] RGB layout) */
int32_t hk_postprocess(hk_ctx* ctx, const hk_postprocess_params* params, int32_t width, int32_t height, const float* src_rgb,
                       const float* depth, float* dst_rgb);

/* ---------------------------------------------------------------------------------------------
 * denoise!(film; config) (src/denoise.jl:301-376): edge-avoiding a-trous wavelet filter.  compute_variance_kernel! (:236-286,
 * 3x3 luminance variance, in-bounds neighbours only) when use_variance, then `iterations` passes of atrous_denoise_kernel!
 * (:136-229; 5x5 B-spline taps at spacing 2^(i-1), clamped to the edge, weights = spatial * colour * normal * depth) ping-ponging
 * between the framebuffer and a scratch buffer.  All buffers are host arrays in Julia [h,w] layout (rgb / normal 3 floats per
 * pixel, depth 1).  dst_rgb receives film.postprocess.  The reference's even passes write INTO film.framebuffer, so after
 * denoise! with iterations >= 2 the framebuffer holds the output of the last even pass: src_after (optional) receives it.
 * ------------------------------------------------------------------------------------------- */
typedef struct hk_denoise_params {
    int32_t iterations;   /* DenoiseConfig defaults (denoise.jl:41-47): 5 */
    float sigma_color;    /* 4 */
    float sigma_normal;   /* 128 */
    float sigma_depth;    /* 1 */
    int32_t use_variance; /* true */
} hk_denoise_params;
int32_t hk_denoise(hk_ctx* ctx, const hk_denoise_params* params, int32_t width, int32_t height, const float* src_rgb, const float* normal,
                   const float* depth, float* dst_rgb, float* src_after);

/* fill_aux_buffers!(film, scene, camera; has_infinite_lights) (src/film.jl:410-483): one primary ray through every pixel
 * centre; albedo = (0.8,0.8,0.8) on a hit else 0, normal = geometric normal of the hit triangle (Raycore's si.core.n is not
 * available here: normalize((v1-v0) x (v2-v0)), un-flipped) else 0, depth = |hit - ray.o|, else +Inf (or 1e30 when the scene
 * has infinite lights).  Outputs are host arrays in Julia [h,w] layout: albedo/normal 3 floats per pixel, depth 1. */
int32_t hk_film_fill_aux(hk_ctx* ctx, hk_scene* scene, const hk_camera* cam, int32_t width, int32_t height, int32_t has_infinite_lights,
                         float* albedo, float* normal, float* depth);

/* point-wise BSDFs of a scene's material `mat_idx` (material-dispatch.jl:23-53; spectral-eval.jl) at uv=(0,0):
   mode 0 = sample_bsdf_spectral(wo, ns, lambda, u, uc, regularize) -> out[10n] = wi3, f4, pdf, is_specular, eta_scale
   mode 1 = evaluate_bsdf_spectral(wo, wi, ns, lambda)              -> out[10n] = f4, pdf, 0...
   wo/wi/ns: 3n floats, lambda: 4n, u: 2n, uc: n. */
int32_t hk_test_bsdf(hk_ctx* ctx, hk_scene* scene, int32_t mode, int32_t mat_idx, int32_t regularize, int32_t n, const float* wo,
                     const float* wi, const float* ns, const float* lambda, const float* u, const float* uc, float* out);
/* point-wise lights (physical-wavefront/lights.jl:39-297, 408-467):
   mode 0 = sample_light_spectral(light_idx_1based, p, lambda, u = in3.xy) -> out[12n] = wi3, pdf, Li4, p_light3, is_delta
   mode 1 = escaped ray along in3 -> out[12n] = Le4 summed over every light, environment pdf, 0... */
int32_t hk_test_light(hk_ctx* ctx, hk_scene* scene, int32_t mode, int32_t light_idx_1based, int32_t n, const float* p3,
                      const float* in3, const float* lambda, float* out);

/* resolve_mix_material (mix-material.jl:96-127, 222-238) for n hit points (p, wo, uv): out_mat = the material index a
   MixMaterial `mat_idx` resolves to (nested mixes followed, textured amount looked up nearest-texel, Q28) */
int32_t hk_test_mix(hk_ctx* ctx, hk_scene* scene, int32_t mat_idx, int32_t n, const float* p3, const float* wo3, const float* uv2,
                    int32_t* out_mat);
/* point-wise media (volpath/media.jl, nanovdb.jl), medium `medium_idx` (0-based), lambda: 4n:
   mode 0 = sample_point at p = a3 (media.jl:1327-1370, 1527-1575; nanovdb.jl:400-469) -> out[13n] = sigma_a4, sigma_s4, Le4, g
   mode 1 = majorant iterator along ray (o = a3, d = b3, t_max) (media.jl:229-340, 625-729; nanovdb.jl:509-554)
            -> out[49n] = number of segments (<= 256), then (t_min, t_max, sigma_maj[lambda 1]) of the first 16 segments
   mode 2 = mode 1 walked the way the tracking kernels do — cells whose majorant is exactly 0 are fast-forwarded without fetching
            the grid — -> out[49n] = number of segments INCLUDING the skipped ones, then the first 16 segments that were not skipped */
int32_t hk_test_medium(hk_ctx* ctx, hk_scene* scene, int32_t mode, int32_t medium_idx, int32_t n, const float* a3, const float* b3,
                       const float* tmax, const float* lambda, float* out);
/* hk_trace_closest through the traversal of the surfaces-only render path (k_trace_lean / k_shadow: while-while rounds with
   per-lane refill and the 16- or 32-entry LDS stack the scene's BVH depth selects).  anyhit != 0: the shadow kernel's
   first-accepted-hit mode, out_prim >= 0 <=> occluded (t / uv then belong to the hit that stopped the ray). */
int32_t hk_test_trace_lean(hk_ctx* ctx, hk_scene* scene, int32_t anyhit, int32_t n, const float* o3, const float* d3, const float* tmax,
                           float* out_t, int32_t* out_prim, float* out_uv2);

/* introspection used by tests/bench */
int32_t hk_scene_bvh_info(hk_scene* scene, int32_t* n_nodes, int32_t* n_leaf_tris, int32_t* max_depth);
int32_t hk_scene_light_bvh_copy(hk_scene* scene, int32_t* n_nodes, float* nodes_out /* 16 floats per node */,
                                uint32_t* bit_trails /* n_lights */);

#ifdef __cplusplus
}
#endif
#endif /* HIKARI_MI355X_H */

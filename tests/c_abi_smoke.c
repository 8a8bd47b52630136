/* c_abi_smoke.c — a plain-C caller of include/hikari_mi355x.h (no Python, no C++, no torch): what a `ccall` user has.
 * Builds a two-triangle matte scene lit by a directional light through the header alone, renders `spp` samples, reads the
 * framebuffer and the raw accumulators back, and writes them to a file that tests/test_c_abi.py compares with the ctypes host.
 *   usage: c_abi_smoke <libhikari_mi355x.so> <data dir> <out file> <width> <height> <spp>
 * The library is dlopen'ed so the test can also prove that a missing GPU is an error and not a silent fallback.
 * Exit codes: 0 ok, 2 no usable GPU (hk_ctx_create failed), 1 anything else. */
#include <dlfcn.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "hikari_mi355x.h"

#define LOAD(name)                                                   \
    name##_t name##_p = (name##_t)dlsym(lib, #name);                 \
    if (!name##_p) {                                                 \
        fprintf(stderr, "missing symbol %s\n", #name);               \
        return 1;                                                    \
    }
typedef int32_t (*hk_ctx_create_t)(int32_t, void*, hk_ctx**);
typedef int32_t (*hk_ctx_destroy_t)(hk_ctx*);
typedef const char* (*hk_last_error_t)(void);
typedef int32_t (*hk_ctx_set_tables_t)(hk_ctx*, const hk_tables*);
typedef int32_t (*hk_scene_create_t)(hk_ctx*, const hk_scene_desc*, hk_scene**);
typedef int32_t (*hk_scene_destroy_t)(hk_scene*);
typedef int32_t (*hk_integrator_create_t)(hk_ctx*, const hk_integrator_params*, hk_integrator**);
typedef int32_t (*hk_integrator_destroy_t)(hk_integrator*);
typedef int32_t (*hk_film_create_t)(hk_ctx*, int32_t, int32_t, int32_t, void*, hk_film**);
typedef int32_t (*hk_film_destroy_t)(hk_film*);
typedef int32_t (*hk_film_clear_t)(hk_film*);
typedef int32_t (*hk_render_t)(hk_ctx*, hk_scene*, hk_integrator*, hk_film*, const hk_camera*, int32_t, int32_t, int32_t);
typedef int32_t (*hk_film_read_rgb_t)(hk_ctx*, hk_film*, float*);
typedef int32_t (*hk_film_read_accum_t)(hk_ctx*, hk_film*, void*);
typedef int32_t (*hk_stats_get_t)(hk_ctx*, hk_stats*);
typedef int32_t (*hk_ctx_set_option_t)(hk_ctx*, const char*, const char*);
typedef int32_t (*hk_ctx_get_option_t)(hk_ctx*, const char*, char*, int32_t);
typedef int32_t (*hk_flush_t)(hk_ctx*);
typedef int32_t (*hk_film_read_rgb_async_t)(hk_ctx*, hk_film*);
typedef int32_t (*hk_film_read_wait_t)(hk_ctx*, hk_film*, float*, const float**);
typedef int32_t (*hk_film_pin_host_t)(hk_film*, float*);
typedef int32_t (*hk_film_unpin_host_t)(hk_film*);

static void* slurp(const char* dir, const char* name, size_t want_bytes) {
    char path[1024];
    snprintf(path, sizeof path, "%s/%s", dir, name);
    FILE* f = fopen(path, "rb");
    if (!f) {
        fprintf(stderr, "cannot open %s\n", path);
        exit(1);
    }
    void* buf = malloc(want_bytes);
    if (fread(buf, 1, want_bytes, f) != want_bytes) {
        fprintf(stderr, "short read of %s\n", path);
        exit(1);
    }
    fclose(f);
    return buf;
}

/* row-major 4x4 helpers for the camera record (camera/perspective.jl:41-94 restated for this fixed camera) */
static void mat_mul(const float* a, const float* b, float* o) {
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            float s = 0.0f;
            for (int k = 0; k < 4; ++k) s += a[4 * i + k] * b[4 * k + j];
            o[4 * i + j] = s;
        }
}

int main(int argc, char** argv) {
    if (argc < 7) {
        fprintf(stderr, "usage: %s lib datadir out w h spp\n", argv[0]);
        return 1;
    }
    const char* datadir = argv[2];
    const int W = atoi(argv[4]), H = atoi(argv[5]), spp = atoi(argv[6]);
    void* lib = dlopen(argv[1], RTLD_NOW);
    if (!lib) {
        fprintf(stderr, "dlopen: %s\n", dlerror());
        return 1;
    }
    LOAD(hk_ctx_create) LOAD(hk_ctx_destroy) LOAD(hk_last_error) LOAD(hk_ctx_set_tables) LOAD(hk_scene_create) LOAD(hk_scene_destroy)
    LOAD(hk_integrator_create) LOAD(hk_integrator_destroy) LOAD(hk_film_create) LOAD(hk_film_destroy) LOAD(hk_film_clear) LOAD(hk_render)
    LOAD(hk_film_read_rgb) LOAD(hk_film_read_accum) LOAD(hk_stats_get)
    LOAD(hk_ctx_set_option) LOAD(hk_ctx_get_option) LOAD(hk_flush) LOAD(hk_film_read_rgb_async) LOAD(hk_film_read_wait)
    LOAD(hk_film_pin_host) LOAD(hk_film_unpin_host)

    hk_ctx* ctx = NULL;
    if (hk_ctx_create_p(0, NULL, &ctx) != HK_OK) {
        fprintf(stderr, "hk_ctx_create: %s\n", hk_last_error_p());
        return 2;
    }
    /* data tables: the files of hikari.jl_amd/data (Sobol matrices, CIE XYZ, the rgb2spec table: int32 res, float scale[res], float coeffs[...]) */
    uint32_t* sobol = (uint32_t*)slurp(datadir, "sobol_matrices.bin", 1024 * 52 * 4);
    float* cie = (float*)slurp(datadir, "cie_xyz.bin", 3 * 471 * 4);
    int32_t res = *(int32_t*)slurp(datadir, "srgb_spectrum_table.dat", 4);
    size_t tab_bytes = 4 + (size_t)res * 4 + (size_t)3 * res * res * res * 3 * 4;
    char* tab = (char*)slurp(datadir, "srgb_spectrum_table.dat", tab_bytes);
    hk_tables T;
    memset(&T, 0, sizeof T);
    T.sobol_matrices = sobol;
    T.sobol_count = 1024 * 52;
    T.rgb2spec_res = res;
    T.cie_x = cie, T.cie_y = cie + 471, T.cie_z = cie + 942;
    T.rgb2spec_scale = (float*)(tab + 4);
    T.rgb2spec_coeffs = (float*)(tab + 4 + (size_t)res * 4);
    if (hk_ctx_set_tables_p(ctx, &T) != HK_OK) {
        fprintf(stderr, "hk_ctx_set_tables: %s\n", hk_last_error_p());
        return 1;
    }
    /* scene: a quad (two triangles) in the z = 0 plane, MatteMaterial(Kd = (0.6, 0.4, 0.2)), DirectionalLight(RGBSpectrum(2), (0.2,-0.3,-1)) */
    float pos[18] = {-1, -1, 0, 1, -1, 0, 1, 1, 0, /**/ -1, -1, 0, 1, 1, 0, -1, 1, 0};
    hk_tri_meta meta[2] = {{0, 1, 0}, {0, 2, 0}};
    hk_material mat;
    memset(&mat, 0, sizeof mat);
    mat.kind = HK_MAT_MATTE;
    for (int k = 0; k < 4; ++k) mat.rgb[k].tex = -1, mat.rgb[k].c[3] = 1.0f;
    for (int k = 0; k < 8; ++k) mat.f[k].tex = -1;
    mat.rgb[0].c[0] = 0.6f, mat.rgb[0].c[1] = 0.4f, mat.rgb[0].c[2] = 0.2f;
    mat.spectrum[0] = mat.spectrum[1] = -1;
    hk_medium_interface mi = {0, -1, -1};
    hk_light light;
    memset(&light, 0, sizeof light);
    light.kind = HK_LIGHT_DIRECTIONAL;
    light.spectrum_kind = HK_SPEC_RGB;
    light.i_rgb[0] = light.i_rgb[1] = light.i_rgb[2] = 2.0f, light.i_rgb[3] = 1.0f;
    light.scale = 1.0f;
    {
        float d[3] = {0.2f, -0.3f, -1.0f};
        float n = 1.0f / sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
        for (int k = 0; k < 3; ++k) light.direction[k] = n * d[k];
    }
    light.Le.tex = -1;
    light.envmap = -1;
    hk_scene_desc D;
    memset(&D, 0, sizeof D);
    D.n_triangles = 2, D.n_materials = 1, D.n_media_interfaces = 1, D.n_lights = 1;
    D.positions = pos, D.meta = meta, D.materials = &mat, D.media_interfaces = &mi, D.lights = &light;
    hk_scene* scene = NULL;
    if (hk_scene_create_p(ctx, &D, &scene) != HK_OK) {
        fprintf(stderr, "hk_scene_create: %s\n", hk_last_error_p());
        return 1;
    }
    /* camera: the record the caller passes on the command line side by side (written by the test: 2 x 16 floats + 10 floats) */
    hk_camera cam;
    {
        char path[1024];
        snprintf(path, sizeof path, "%s.camera", argv[3]);
        FILE* f = fopen(path, "rb");
        if (!f || fread(&cam, sizeof cam, 1, f) != 1) {
            fprintf(stderr, "cannot read the camera record %s\n", path);
            return 1;
        }
        fclose(f);
        float id[16];
        mat_mul(cam.camera_to_world, cam.camera_to_world, id); /* (touch the matrices: a caller owns this memory) */
    }
    hk_integrator_params P;
    memset(&P, 0, sizeof P);
    P.max_depth = 3, P.samples_per_pixel = spp, P.russian_roulette_depth = 3, P.regularize = 1, P.max_component_value = 10.0f;
    P.filter_type = HK_FILTER_GAUSSIAN, P.filter_radius[0] = P.filter_radius[1] = 1.5f, P.filter_param1 = 0.5f;
    hk_integrator* integ = NULL;
    hk_film* film = NULL;
    if (hk_integrator_create_p(ctx, &P, &integ) != HK_OK || hk_film_create_p(ctx, W, H, 0, NULL, &film) != HK_OK) {
        fprintf(stderr, "create: %s\n", hk_last_error_p());
        return 1;
    }
    hk_film_clear_p(film);
    if (hk_render_p(ctx, scene, integ, film, &cam, 1, spp, 1) != HK_OK) {
        fprintf(stderr, "hk_render: %s\n", hk_last_error_p());
        return 1;
    }
    float* rgb = (float*)malloc((size_t)W * H * 3 * 4);
    float* acc = (float*)malloc((size_t)W * H * 4 * 4);
    if (hk_film_read_rgb_p(ctx, film, rgb) != HK_OK || hk_film_read_accum_p(ctx, film, acc) != HK_OK) {
        fprintf(stderr, "read: %s\n", hk_last_error_p());
        return 1;
    }
    hk_stats st;
    hk_stats_get_p(ctx, &st);
    FILE* out = fopen(argv[3], "wb");
    fwrite(rgb, 4, (size_t)W * H * 3, out);
    fwrite(acc, 4, (size_t)W * H * 4, out);
    fclose(out);
    printf("c_abi_smoke ok: %d x %d, %d spp, rays %llu + %llu\n", W, H, spp, (unsigned long long)st.rays_closest, (unsigned long long)st.rays_shadow);
    /* error behaviour: bad arguments are status codes with a message, never a crash */
    if (hk_render_p(ctx, scene, integ, film, &cam, 0, 1, 1) != HK_ERR_INVALID || !hk_last_error_p()[0]) return 1;
    /* knobs live in the context (the environment is read once, by hk_ctx_create): set, read back, reset; unknown names are refused */
    {
        char buf[16];
        if (hk_ctx_set_option_p(ctx, "HK_BATCH_PATHS_M", "0") != HK_OK || hk_ctx_get_option_p(ctx, "HK_BATCH_PATHS_M", buf, 16) != 1 || buf[0] != '0') return 1;
        if (hk_ctx_set_option_p(ctx, "HK_BATCH_PATHS_M", NULL) != HK_OK || hk_ctx_get_option_p(ctx, "HK_BATCH_PATHS_M", buf, 16) != HK_UNSET || buf[0] != 0) return 1;   /* no value: HK_UNSET, not an error code */
        if (hk_ctx_get_option_p(ctx, "HK_NOT_A_KNOB", buf, 16) != HK_ERR_INVALID) return 1;
        if (hk_ctx_set_option_p(ctx, "HK_NOT_A_KNOB", "1") != HK_ERR_INVALID) return 1;
    }
    /* the interactive loop of a viewer: one more sample per call, the frame of call i - 1 collected while call i renders; the last frame
       through the pair equals a synchronous read of the same film */
    {
        const float* frame = NULL;
        float* rgb2 = (float*)malloc((size_t)W * H * 3 * 4);
        for (int i = 0; i < 3; ++i) {
            if (hk_render_p(ctx, scene, integ, film, &cam, spp + 1 + i, 1, 1) != HK_OK || hk_flush_p(ctx) != HK_OK) return 1;
            if (i > 0 && hk_film_read_wait_p(ctx, film, NULL, &frame) != HK_OK) return 1;
            if (hk_film_read_rgb_async_p(ctx, film) != HK_OK) return 1;
        }
        if (hk_film_read_wait_p(ctx, film, NULL, &frame) != HK_OK || !frame) return 1;
        if (hk_film_read_rgb_p(ctx, film, rgb2) != HK_OK || memcmp(frame, rgb2, (size_t)W * H * 3 * 4) != 0) {
            fprintf(stderr, "asynchronous and synchronous frame differ\n");
            return 1;
        }
        /* a viewer's ONE frame buffer, named to the library: the frame is copied straight into it (no staging memcpy); a malloc / free per
           frame WITHOUT naming the buffer must be just as safe — the library registers nothing on its own */
        float* pinned = (float*)malloc((size_t)W * H * 3 * 4);
        if (hk_film_pin_host_p(film, pinned) != HK_OK || hk_film_read_rgb_p(ctx, film, pinned) != HK_OK || memcmp(pinned, rgb2, (size_t)W * H * 3 * 4) != 0) return 1;
        if (hk_film_read_rgb_p(ctx, film, pinned) != HK_OK || hk_film_unpin_host_p(film) != HK_OK) return 1;
        free(pinned);
        for (int i = 0; i < 3; ++i) {
            float* tmp = (float*)malloc((size_t)W * H * 3 * 4);
            if (hk_film_read_rgb_p(ctx, film, tmp) != HK_OK || memcmp(tmp, rgb2, (size_t)W * H * 3 * 4) != 0) return 1;
            free(tmp);
        }
        free(rgb2);
    }
    hk_film_destroy_p(film);
    hk_integrator_destroy_p(integ);
    hk_scene_destroy_p(scene);
    hk_ctx_destroy_p(ctx);
    return 0;
}

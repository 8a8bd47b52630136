// hko_media.h — CPU ORACLE (test infrastructure): participating media.
// Follows:
//   HG phase function                     src/integrators/volpath/media.jl:16-76
//   MajorantGrid / DDA iterator           src/integrators/volpath/media.jl:178-205, 229-340, 625-729
//   HomogeneousMedium                     src/integrators/volpath/media.jl:762-851
//   GridMedium (density, sample, DDA)     src/integrators/volpath/media.jl:873-935, 1527-1696
//   ray_bounds_intersect                  src/integrators/volpath/media.jl:1698-1734
//   NanoVDB tree walk / trilinear sample  src/integrators/volpath/nanovdb.jl:198-242, 252-299, 315-388, 400-469, 477-554
//   LCG                                   src/integrators/volpath/delta-tracking.jl:18-58
//   ratio tracking                        src/integrators/volpath/intersection.jl:422-542
#pragma once
#include "hikari_mi355x.h"
#include "hko_sampler.h"
#include "hko_spectral.h"

namespace hko {

inline float hg_p(float g, float cos_t) {
    float g2 = g * g;
    float denom = 1.0f + g2 - 2.0f * g * cos_t;
    return (1.0f - g2) / (4.0f * PI_F * denom * std::sqrt(denom));
}
inline void coordinate_system_m(V3 n, V3& tangent, V3& bitangent) {  // spectral-eval.jl:3514-3533
    if (std::fabs(n.x) > std::fabs(n.y)) {
        float inv_len = 1.0f / std::sqrt(n.x * n.x + n.z * n.z);
        tangent = V3(n.z * inv_len, 0.0f, -n.x * inv_len);
    } else {
        float inv_len = 1.0f / std::sqrt(n.y * n.y + n.z * n.z);
        tangent = V3(0.0f, n.z * inv_len, -n.y * inv_len);
    }
    bitangent = cross(n, tangent);
}
inline V3 sample_hg(float g, V3 wo, V2 u, float& pdf) {  // media.jl:51-72
    float cos_t;
    if (std::fabs(g) < 1e-3f)
        cos_t = 1.0f - 2.0f * u.x;
    else {
        float g2 = g * g;
        float sqr = (1.0f - g2) / (1.0f - g + 2.0f * g * u.x);
        cos_t = clampf((1.0f + g2 - sqr * sqr) / (2.0f * g), -1.0f, 1.0f);
    }
    float sin_t = std::sqrt(maxf(0.0f, 1.0f - cos_t * cos_t));
    float phi = 2.0f * PI_F * u.y;
    V3 t1, t2;
    coordinate_system_m(-wo, t1, t2);
    float sphi, cphi;
    jl_sincos(phi, sphi, cphi);
    V3 wi = sin_t * cphi * t1 + sin_t * sphi * t2 + cos_t * (-wo);
    wi = normalize(wi);
    pdf = hg_p(g, cos_t);
    return wi;
}

// LCG (delta-tracking.jl:28-58)
inline uint64_t lcg_init(V3 o, V3 d, float t_max) {
    uint64_t ox = f2u(o.x), oy = f2u(o.y), oz = f2u(o.z), tm = f2u(t_max);
    uint64_t dx = f2u(d.x), dy = f2u(d.y), dz = f2u(d.z);
    uint64_t s1 = mix_bits(ox ^ (oy << 16) ^ (oz << 32) ^ tm);
    uint64_t s2 = mix_bits(dx ^ (dy << 16) ^ (dz << 32));
    return s1 ^ s2;
}
inline float lcg_next(uint64_t& state) {
    state = state * 0x5DEECE66Dull + 11ull;
    float r = (float)(uint32_t)(state >> 32) * 2.3283064365386963e-10f;
    const float lim = 1.0f - 1.1920929e-7f;
    return r < lim ? r : lim;
}

struct MediumProperties {
    Spec sigma_a, sigma_s, Le;
    float g = 0.0f;
};
struct MajorantSegment {
    float t_min = 0, t_max = 0;
    Spec sigma_maj;
};
// RayMajorantIterator (media.jl:517-560): mode 0 exhausted, 1 homogeneous, 2 DDA
struct MajorantIter {
    int32_t mode = 0;
    Spec sigma_t;
    float t_min = INF_F, t_max = -INF_F;
    bool hom_called = true;
    const float* grid = nullptr;
    int32_t res[3] = {1, 1, 1};
    float next_t[3] = {0, 0, 0}, delta_t[3] = {0, 0, 0};
    int32_t step[3] = {0, 0, 0}, limit[3] = {0, 0, 0}, voxel[3] = {0, 0, 0};
};

struct MediaCtx {
    const hk_medium* media = nullptr;
    int32_t n = 0;
    const RGB2SpecTable* table = nullptr;
};
inline void init_media(MediaCtx& m, const hk_scene_desc* d, const RGB2SpecTable* t) {
    m.media = d->media;
    m.n = d->n_media;
    m.table = t;
}

inline void ray_bounds_intersect(V3 o, V3 d, const float* bmin, const float* bmax, float& t_enter, float& t_exit) {
    float inv[3], t0[3], t1[3];
    for (int k = 0; k < 3; ++k) {
        float dk = d[k];
        inv[k] = std::fabs(dk) > 1e-10f ? 1.0f / dk : (dk >= 0 ? INF_F : -INF_F);
        t0[k] = (bmin[k] - o[k]) * inv[k];
        t1[k] = (bmax[k] - o[k]) * inv[k];
        if (t0[k] > t1[k]) std::swap(t0[k], t1[k]);
    }
    t_enter = maxf(maxf(t0[0], t0[1]), t0[2]);
    t_exit = minf(minf(t1[0], t1[1]), t1[2]);
}

// create_dda_iterator (media.jl:229-340)
inline MajorantIter create_dda(const float* grid, const int32_t res[3], const float* bmin, const float* bmax, V3 ro, V3 rd, float t_min, float t_max, const Spec& sigma_t) {
    MajorantIter it;
    it.grid = grid;
    float gi[3], gd[3];
    for (int k = 0; k < 3; ++k) {
        it.res[k] = res[k];
        float diag = bmax[k] - bmin[k];
        float go = (ro[k] - bmin[k]) / diag;
        float inv_diag = std::fabs(diag) > 1e-10f ? 1.0f / diag : 0.0f;
        gd[k] = rd[k] * inv_diag;
        gi[k] = go + gd[k] * t_min;
        int32_t v = clampi(floor_int32(gi[k] * (float)res[k]), 0, res[k] - 1);
        it.voxel[k] = v;
        it.delta_t[k] = std::fabs(gd[k]) > 1e-10f ? 1.0f / (std::fabs(gd[k]) * (float)res[k]) : INF_F;
        if (gd[k] >= 0.0f) {
            float nvp = (float)(v + 1) / (float)res[k];
            it.next_t[k] = gd[k] > 1e-10f ? t_min + (nvp - gi[k]) / gd[k] : INF_F;
            it.step[k] = 1;
            it.limit[k] = res[k];
        } else {
            float nvp = (float)v / (float)res[k];
            it.next_t[k] = gd[k] < -1e-10f ? t_min + (nvp - gi[k]) / gd[k] : INF_F;
            it.step[k] = -1;
            it.limit[k] = -1;
        }
    }
    it.sigma_t = sigma_t;
    it.t_min = t_min;
    it.t_max = t_max;
    it.hom_called = false;
    it.mode = t_min >= t_max ? 0 : 2;
    return it;
}
// ray_majorant_next (media.jl:625-729)
inline bool majorant_next(MajorantIter& it, MajorantSegment& seg) {
    if (it.mode == 0) return false;
    if (it.mode == 1) {
        if (it.hom_called || it.t_min >= it.t_max) {
            it.mode = 0;
            it.hom_called = true;
            return false;
        }
        seg.t_min = it.t_min;
        seg.t_max = it.t_max;
        seg.sigma_maj = it.sigma_t;
        it.hom_called = true;
        return true;
    }
    if (it.t_min >= it.t_max) {
        it.mode = 0;
        return false;
    }
    float tx = it.next_t[0], ty = it.next_t[1], tz = it.next_t[2];
    int axis = (tx < ty) ? ((tx < tz) ? 0 : 2) : ((ty < tz) ? 1 : 2);
    float seg_t_max = minf(it.next_t[axis], it.t_max);
    float rho = it.grid[it.voxel[0] + it.res[0] * (it.voxel[1] + it.res[1] * it.voxel[2])];
    seg.t_min = it.t_min;
    seg.t_max = seg_t_max;
    seg.sigma_maj = it.sigma_t * rho;
    it.t_min = seg_t_max;
    it.voxel[axis] += it.step[axis];
    it.next_t[axis] += it.delta_t[axis];
    if (it.voxel[0] == it.limit[0] || it.voxel[1] == it.limit[1] || it.voxel[2] == it.limit[2]) {
        it.mode = 0;
        it.t_min = it.t_max;
    }
    return true;
}

inline Spec medium_uplift(const MediaCtx& mc, const float rgb[4], const Wavelengths& w) { return uplift_rgb_unbounded(*mc.table, RGBA(rgb[0], rgb[1], rgb[2], rgb[3]), w); }

// ---- NanoVDB (nanovdb.jl) : offsets are 1-based byte positions like the Julia fields ----
inline float nv_f32(const uint8_t* b, int64_t off1) {
    float v;
    std::memcpy(&v, b + (((off1 - 1) >> 2) << 2), 4);
    return v;
}
inline int64_t nv_i64(const uint8_t* b, int64_t off1) {
    int64_t v;
    std::memcpy(&v, b + (((off1 - 1) >> 3) << 3), 8);
    return v;
}
inline uint64_t nv_u64(const uint8_t* b, int64_t off1) { return (uint64_t)nv_i64(b, off1); }
inline bool nv_mask(const uint8_t* b, int64_t mask_off1, int32_t n) { return ((b[mask_off1 - 1 + (n >> 3)] >> (n & 7)) & 1) != 0; }
inline float nanovdb_get_value(const hk_medium& m, int32_t x, int32_t y, int32_t z) {  // nanovdb.jl:315-388
    const uint8_t* b = m.nvdb_bytes;
    int64_t root = m.root_offset_1based;
    uint32_t xu = (uint32_t)x, yu = (uint32_t)y, zu = (uint32_t)z;
    uint64_t key = (uint64_t)((zu >> 12) & 0x1fffff) | ((uint64_t)((yu >> 12) & 0x1fffff) << 21) | ((uint64_t)((xu >> 12) & 0x1fffff) << 42);
    int64_t tile = 0;
    bool found = false;
    int64_t tile_base = root + 64;
    for (int32_t i = 0; i < m.root_table_size; ++i) {
        int64_t t_off = tile_base + (int64_t)i * 32;
        if (nv_u64(b, t_off) == key) {
            found = true;
            tile = t_off;
            break;
        }
    }
    if (!found) return nv_f32(b, root + 28);
    int64_t child = nv_i64(b, tile + 8);
    if (child == 0) return nv_f32(b, tile + 20);
    int64_t upper = root + child;
    int32_t n_upper = (int32_t)(((xu >> 7) & 31) << 10) | (int32_t)(((yu >> 7) & 31) << 5) | (int32_t)((zu >> 7) & 31);
    if (!nv_mask(b, upper + 4128, n_upper)) return nv_f32(b, upper + 8256 + (int64_t)n_upper * 8);
    int64_t lower = upper + nv_i64(b, upper + 8256 + (int64_t)n_upper * 8);
    int32_t n_lower = (int32_t)(((xu >> 3) & 15) << 8) | (int32_t)(((yu >> 3) & 15) << 4) | (int32_t)((zu >> 3) & 15);
    if (!nv_mask(b, lower + 544, n_lower)) return nv_f32(b, lower + 1088 + (int64_t)n_lower * 8);
    int64_t leaf = lower + nv_i64(b, lower + 1088 + (int64_t)n_lower * 8);
    int32_t n_leaf = ((x & 7) << 6) | ((y & 7) << 3) | (z & 7);
    return nv_f32(b, leaf + 96 + (int64_t)n_leaf * 4);
}
inline float sample_nanovdb_density(const hk_medium& m, V3 p) {  // nanovdb.jl:400-469
    float px = p.x - m.vec[0], py = p.y - m.vec[1], pz = p.z - m.vec[2];
    float fxi = m.inv_mat[0] * px + m.inv_mat[1] * py + m.inv_mat[2] * pz;
    float fyi = m.inv_mat[3] * px + m.inv_mat[4] * py + m.inv_mat[5] * pz;
    float fzi = m.inv_mat[6] * px + m.inv_mat[7] * py + m.inv_mat[8] * pz;
    int32_t ix = floor_int32(fxi), iy = floor_int32(fyi), iz = floor_int32(fzi);
    float fx = fxi - (float)ix, fy = fyi - (float)iy, fz = fzi - (float)iz;
    float v000 = nanovdb_get_value(m, ix, iy, iz), v001 = nanovdb_get_value(m, ix, iy, iz + 1);
    float v010 = nanovdb_get_value(m, ix, iy + 1, iz), v011 = nanovdb_get_value(m, ix, iy + 1, iz + 1);
    float v100 = nanovdb_get_value(m, ix + 1, iy, iz), v101 = nanovdb_get_value(m, ix + 1, iy, iz + 1);
    float v110 = nanovdb_get_value(m, ix + 1, iy + 1, iz), v111 = nanovdb_get_value(m, ix + 1, iy + 1, iz + 1);
    float fx1 = 1.0f - fx, fy1 = 1.0f - fy, fz1 = 1.0f - fz;
    float v00 = v000 * fz1 + v001 * fz, v01 = v010 * fz1 + v011 * fz;
    float v10 = v100 * fz1 + v101 * fz, v11 = v110 * fz1 + v111 * fz;
    float v0 = v00 * fy1 + v01 * fy, v1 = v10 * fy1 + v11 * fy;
    return v0 * fx1 + v1 * fx;
}

// GridMedium density (media.jl:1527-1575): density[x,y,z] Julia layout, 1-based
inline float grid_density(const hk_medium& m, int32_t ix, int32_t iy, int32_t iz) {
    return m.density[(size_t)(ix - 1) + (size_t)m.res[0] * ((size_t)(iy - 1) + (size_t)m.res[1] * (size_t)(iz - 1))];
}
inline float sample_grid_density(const hk_medium& m, V3 pm) {
    float pn[3];
    for (int k = 0; k < 3; ++k) pn[k] = (pm[k] - m.bounds_min[k]) / (m.bounds_max[k] - m.bounds_min[k]);
    if (pn[0] < 0.0f || pn[1] < 0.0f || pn[2] < 0.0f || pn[0] > 1.0f || pn[1] > 1.0f || pn[2] > 1.0f) return 0.0f;
    int32_t nx = m.res[0], ny = m.res[1], nz = m.res[2];
    float gx = pn[0] * (float)nx + 0.5f, gy = pn[1] * (float)ny + 0.5f, gz = pn[2] * (float)nz + 0.5f;
    int32_t ix = clampi(floor_int32(gx), 1, nx - 1), iy = clampi(floor_int32(gy), 1, ny - 1), iz = clampi(floor_int32(gz), 1, nz - 1);
    float fx = clampf(gx - (float)ix, 0.0f, 1.0f), fy = clampf(gy - (float)iy, 0.0f, 1.0f), fz = clampf(gz - (float)iz, 0.0f, 1.0f);
    float d000 = grid_density(m, ix, iy, iz), d100 = grid_density(m, ix + 1, iy, iz), d010 = grid_density(m, ix, iy + 1, iz), d110 = grid_density(m, ix + 1, iy + 1, iz);
    float d001 = grid_density(m, ix, iy, iz + 1), d101 = grid_density(m, ix + 1, iy, iz + 1), d011 = grid_density(m, ix, iy + 1, iz + 1),
          d111 = grid_density(m, ix + 1, iy + 1, iz + 1);
    float fx1 = 1.0f - fx;
    float d00 = d000 * fx1 + d100 * fx, d10 = d010 * fx1 + d110 * fx, d01 = d001 * fx1 + d101 * fx, d11 = d011 * fx1 + d111 * fx;
    float fy1 = 1.0f - fy;
    float d0 = d00 * fy1 + d10 * fy, d1 = d01 * fy1 + d11 * fy;
    return d0 * (1.0f - fz) + d1 * fz;
}
// RGBGridMedium grids (media.jl:1252-1324): RGBSpectrum[x,y,z] Julia layout, 4 floats per voxel, cell-centred trilinear
inline RGBA rgb_grid_at(const float* g, const hk_medium& m, int32_t ix, int32_t iy, int32_t iz) {
    const float* p = g + 4 * ((size_t)(ix - 1) + (size_t)m.res[0] * ((size_t)(iy - 1) + (size_t)m.res[1] * (size_t)(iz - 1)));
    return RGBA(p[0], p[1], p[2], p[3]);
}
inline RGBA rgba_lerp(const RGBA& a, float wa, const RGBA& b, float wb) {  // a*wa + b*wb
    return RGBA(a.c[0] * wa + b.c[0] * wb, a.c[1] * wa + b.c[1] * wb, a.c[2] * wa + b.c[2] * wb, a.c[3] * wa + b.c[3] * wb);
}
inline RGBA sample_rgb_grid(const float* g, const hk_medium& m, const float pn[3]) {
    if (pn[0] < 0.0f || pn[1] < 0.0f || pn[2] < 0.0f || pn[0] > 1.0f || pn[1] > 1.0f || pn[2] > 1.0f) return RGBA(0, 0, 0, 0);
    int32_t nx = m.res[0], ny = m.res[1], nz = m.res[2];
    float gx = pn[0] * (float)nx + 0.5f, gy = pn[1] * (float)ny + 0.5f, gz = pn[2] * (float)nz + 0.5f;
    int32_t ix = clampi(floor_int32(gx), 1, nx - 1), iy = clampi(floor_int32(gy), 1, ny - 1), iz = clampi(floor_int32(gz), 1, nz - 1);
    float fx = clampf(gx - (float)ix, 0.0f, 1.0f), fy = clampf(gy - (float)iy, 0.0f, 1.0f), fz = clampf(gz - (float)iz, 0.0f, 1.0f);
    float fx1 = 1.0f - fx, fy1 = 1.0f - fy;
    RGBA c00 = rgba_lerp(rgb_grid_at(g, m, ix, iy, iz), fx1, rgb_grid_at(g, m, ix + 1, iy, iz), fx);
    RGBA c10 = rgba_lerp(rgb_grid_at(g, m, ix, iy + 1, iz), fx1, rgb_grid_at(g, m, ix + 1, iy + 1, iz), fx);
    RGBA c01 = rgba_lerp(rgb_grid_at(g, m, ix, iy, iz + 1), fx1, rgb_grid_at(g, m, ix + 1, iy, iz + 1), fx);
    RGBA c11 = rgba_lerp(rgb_grid_at(g, m, ix, iy + 1, iz + 1), fx1, rgb_grid_at(g, m, ix + 1, iy + 1, iz + 1), fx);
    RGBA c0 = rgba_lerp(c00, fy1, c10, fy), c1 = rgba_lerp(c01, fy1, c11, fy);
    return rgba_lerp(c0, 1.0f - fz, c1, fz);
}
inline V3 xform_affine_point(const float* M, V3 p) {
    return V3(M[0] * p.x + M[1] * p.y + M[2] * p.z + M[3], M[4] * p.x + M[5] * p.y + M[6] * p.z + M[7], M[8] * p.x + M[9] * p.y + M[10] * p.z + M[11]);
}
inline V3 xform_affine_dir(const float* M, V3 d) {
    return V3(M[0] * d.x + M[1] * d.y + M[2] * d.z, M[4] * d.x + M[5] * d.y + M[6] * d.z, M[8] * d.x + M[9] * d.y + M[10] * d.z);
}

// sample_point dispatch (media.jl:781-793, 1597-1623; nanovdb.jl:477-492)
inline MediumProperties sample_point(const MediaCtx& mc, int32_t idx, V3 p, const Wavelengths& w) {
    const hk_medium& m = mc.media[idx];
    MediumProperties mp;
    mp.g = m.g;
    switch (m.kind) {
        case HK_MEDIUM_HOMOGENEOUS:
            mp.sigma_a = medium_uplift(mc, m.sigma_a, w);
            mp.sigma_s = medium_uplift(mc, m.sigma_s, w);
            mp.Le = medium_uplift(mc, m.Le, w);
            return mp;
        case HK_MEDIUM_GRID: {
            float d = sample_grid_density(m, xform_affine_point(m.render_to_medium, p));
            mp.sigma_a = medium_uplift(mc, m.sigma_a, w) * d;
            mp.sigma_s = medium_uplift(mc, m.sigma_s, w) * d;
            mp.Le = Spec(0.0f);
            return mp;
        }
        case HK_MEDIUM_RGB_GRID: {  // media.jl:1327-1370
            V3 pm = xform_affine_point(m.render_to_medium, p);
            float pn[3];
            for (int k = 0; k < 3; ++k) pn[k] = (pm[k] - m.bounds_min[k]) / (m.bounds_max[k] - m.bounds_min[k]);
            RGBA a = m.sigma_a_grid ? sample_rgb_grid(m.sigma_a_grid, m, pn) : RGBA(1, 1, 1, 1);
            RGBA sc = m.sigma_s_grid ? sample_rgb_grid(m.sigma_s_grid, m, pn) : RGBA(1, 1, 1, 1);
            mp.sigma_a = uplift_rgb_unbounded(*mc.table, a, w) * m.sigma_scale;
            mp.sigma_s = uplift_rgb_unbounded(*mc.table, sc, w) * m.sigma_scale;
            mp.Le = Spec(0.0f);
            if (m.Le_grid && m.Le_scale > 0.0f) mp.Le = uplift_rgb_unbounded(*mc.table, sample_rgb_grid(m.Le_grid, m, pn), w) * m.Le_scale;
            return mp;
        }
        case HK_MEDIUM_NANOVDB: {
            float d = sample_nanovdb_density(m, p);
            mp.sigma_a = medium_uplift(mc, m.sigma_a, w) * d;
            mp.sigma_s = medium_uplift(mc, m.sigma_s, w) * d;
            mp.Le = Spec(0.0f);
            return mp;
        }
        default: return mp;
    }
}
// create_majorant_iterator dispatch (media.jl:810-851, 1625-1696; nanovdb.jl:509-554)
inline MajorantIter create_majorant_iterator(const MediaCtx& mc, int32_t idx, V3 ro, V3 rd, float t_max, const Wavelengths& w) {
    const hk_medium& m = mc.media[idx];
    MajorantIter it;
    Spec sa = medium_uplift(mc, m.sigma_a, w), ss = medium_uplift(mc, m.sigma_s, w);
    if (m.kind == HK_MEDIUM_HOMOGENEOUS) {
        it.mode = (0.0f >= t_max) ? 0 : 1;
        it.sigma_t = sa + ss;
        it.t_min = 0.0f;
        it.t_max = t_max;
        it.hom_called = false;
        return it;
    }
    V3 o = ro, d = rd;
    if (m.kind == HK_MEDIUM_GRID || m.kind == HK_MEDIUM_RGB_GRID) {
        o = xform_affine_point(m.render_to_medium, ro);
        d = xform_affine_dir(m.render_to_medium, rd);
        float len_sq = d.x * d.x + d.y * d.y + d.z * d.z;
        if (len_sq < 1e-20f) return it;
    }
    float t_enter, t_exit;
    ray_bounds_intersect(o, d, m.bounds_min, m.bounds_max, t_enter, t_exit);
    t_enter = maxf(t_enter, 0.0f);
    t_exit = minf(t_exit, t_max);
    if (t_enter >= t_exit) return it;
    // RGBGridMedium: unit sigma_t, the scale lives in the majorant grid (media.jl:1408-1420)
    return create_dda(m.majorant, m.majorant_res, m.bounds_min, m.bounds_max, o, d, t_enter, t_exit, m.kind == HK_MEDIUM_RGB_GRID ? Spec(1.0f) : sa + ss);
}

// compute_transmittance_ratio_tracking + _ratio_tracking_dda (intersection.jl:422-542)
inline void transmittance_ratio_tracking(const MediaCtx& mc, int32_t medium_idx, V3 origin, V3 dir, float t_max, const Wavelengths& lambda, Spec& T_ray, Spec& r_u, Spec& r_l,
                                         uint64_t& collisions) {
    T_ray = Spec(1.0f);
    r_u = Spec(1.0f);
    r_l = Spec(1.0f);
    MajorantIter it = create_majorant_iterator(mc, medium_idx, origin, dir, t_max, lambda);
    PCG32 rng = pcg32_init(pbrt_hash(origin), pbrt_hash(dir));
    for (int s = 0; s < 256; ++s) {
        MajorantSegment seg;
        if (!majorant_next(it, seg)) break;
        Spec sm = seg.sigma_maj;
        float sm0 = sm[0];
        if (sm0 < 1e-10f) continue;
        float t = seg.t_min;
        for (int k = 0; k < 100; ++k) {
            float u = pcg32_uniform_f32(rng);
            float dt = -std::log(maxf(1e-10f, 1.0f - u)) / sm0;
            float t_sample = t + dt;
            if (t_sample >= seg.t_max) {
                float dt_remain = seg.t_max - t;
                Spec T_maj = exp(-dt_remain * sm);
                float T0 = T_maj[0];
                if (T0 > 1e-10f) {
                    T_ray = T_ray * T_maj / T0;
                    r_l = r_l * T_maj / T0;
                    r_u = r_u * T_maj / T0;
                }
                break;
            }
            ++collisions;
            V3 p = origin + dir * t_sample;
            MediumProperties mp = sample_point(mc, medium_idx, p, lambda);
            Spec sn = sm - mp.sigma_a - mp.sigma_s;
            sn = Spec(maxf(sn[0], 0.0f), maxf(sn[1], 0.0f), maxf(sn[2], 0.0f), maxf(sn[3], 0.0f));
            Spec T_maj = exp(-dt * sm);
            float pr = T_maj[0] * sm0;
            if (pr > 1e-10f) {
                T_ray = T_ray * T_maj * sn / pr;
                r_l = r_l * T_maj * sm / pr;
                r_u = r_u * T_maj * sn / pr;
            } else {
                T_ray = Spec(0.0f);
                return;
            }
            Spec Tr_est = T_ray / maxf(1e-10f, average(r_l + r_u));
            if (max_component(Tr_est) < 0.05f) {
                float q = 0.75f;
                float rr = pcg32_uniform_f32(rng);
                if (rr < q) {
                    T_ray = Spec(0.0f);
                    return;
                }
                T_ray = T_ray / (1.0f - q);
            }
            if (is_black(T_ray)) return;
            t = t_sample;
        }
        if (is_black(T_ray)) break;
    }
}

}  // namespace hko

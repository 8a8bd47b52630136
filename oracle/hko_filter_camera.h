// hko_filter_camera.h — CPU ORACLE (test infrastructure): pixel filter sampling and camera rays.
// Follows:
//   filters (evaluate)                 src/filter.jl:40-316
//   GPUFilterSamplerData (tabulation)  src/filter.jl:636-725
//   filter_sample_tabulated & friends  src/filter.jl:733-953
//   sample_tent                        src/filter.jl:101-118
//   PerspectiveCamera.generate_ray     src/camera/perspective.jl:95-128
#pragma once
#include <vector>

#include "hikari_mi355x.h"
#include "hko_sampler.h"

namespace hko {

struct FilterParams {
    int32_t type;
    float rx, ry;
    float p1, p2;
    float exp_x, exp_y;
};
inline float gaussian_1d(float x, float sigma) { return std::exp(-(x * x) / (2.0f * (sigma * sigma))); }
inline float mitchell_1d(float x, float B, float C) {
    x = std::fabs(x);
    if (x <= 1.0f)
        return ((12.0f - 9.0f * B - 6.0f * C) * (x * x * x) + (-18.0f + 12.0f * B + 6.0f * C) * (x * x) + (6.0f - 2.0f * B)) / 6.0f;
    else if (x <= 2.0f)
        return ((-B - 6.0f * C) * (x * x * x) + (6.0f * B + 30.0f * C) * (x * x) + (-12.0f * B - 48.0f * C) * x + (8.0f * B + 24.0f * C)) / 6.0f;
    return 0.0f;
}
inline float sinc_func(float x) {
    x = std::fabs(x);
    if (x < 1e-5f) return 1.0f;
    x *= PI_F;
    return std::sin(x) / x;
}
inline float windowed_sinc(float x, float r, float tau) {
    x = std::fabs(x);
    if (x > r) return 0.0f;
    return sinc_func(x) * sinc_func(x / tau);
}
inline float filter_evaluate(const FilterParams& f, float px, float py) {
    switch (f.type) {
        case HK_FILTER_BOX: return (std::fabs(px) <= f.rx && std::fabs(py) <= f.ry) ? 1.0f : 0.0f;
        case HK_FILTER_TRIANGLE: return maxf(0.0f, f.rx - std::fabs(px)) * maxf(0.0f, f.ry - std::fabs(py));
        case HK_FILTER_GAUSSIAN: {
            float gx = maxf(0.0f, gaussian_1d(px, f.p1) - f.exp_x);
            float gy = maxf(0.0f, gaussian_1d(py, f.p1) - f.exp_y);
            return gx * gy;
        }
        case HK_FILTER_MITCHELL: return mitchell_1d(2.0f * px / f.rx, f.p1, f.p2) * mitchell_1d(2.0f * py / f.ry, f.p1, f.p2);
        case HK_FILTER_LANCZOS: return windowed_sinc(px, f.rx, f.p1) * windowed_sinc(py, f.ry, f.p1);
    }
    return 0.0f;
}
inline FilterParams make_filter_params(const hk_integrator_params& p) {
    FilterParams f;
    f.type = p.filter_type;
    f.rx = p.filter_radius[0];
    f.ry = p.filter_radius[1];
    f.p1 = p.filter_param1;
    f.p2 = p.filter_param2;
    f.exp_x = f.exp_y = 0.0f;
    if (f.type == HK_FILTER_GAUSSIAN) {
        f.exp_x = gaussian_1d(f.rx, f.p1);
        f.exp_y = gaussian_1d(f.ry, f.p1);
    }
    return f;
}

// GPUFilterSamplerData (filter.jl:611-725).  func[iy,ix] stored row = iy.
struct FilterSampler {
    bool valid = false;
    int32_t nx = 0, ny = 0;
    std::vector<float> func;             // ny*nx, index iy*nx+ix (0-based)
    std::vector<float> marginal_cdf;     // ny+1
    std::vector<float> marginal_func;    // ny
    std::vector<float> conditional_cdf;  // ny*(nx+1)
    float dmin_x, dmin_y, dmax_x, dmax_y;
    float func_integral;
};
inline FilterSampler build_filter_sampler(const FilterParams& f) {
    FilterSampler s;
    if (f.type == HK_FILTER_BOX || f.type == HK_FILTER_TRIANGLE) return s;
    s.valid = true;
    int32_t nx = (int32_t)std::ceil(32 * f.rx), ny = (int32_t)std::ceil(32 * f.ry);
    if (nx < 8) nx = 8;
    if (ny < 8) ny = 8;
    s.nx = nx;
    s.ny = ny;
    s.dmin_x = -f.rx;
    s.dmin_y = -f.ry;
    s.dmax_x = f.rx;
    s.dmax_y = f.ry;
    float dx = (s.dmax_x - s.dmin_x) / (float)nx, dy = (s.dmax_y - s.dmin_y) / (float)ny;
    s.func.assign((size_t)nx * ny, 0.0f);
    for (int iy = 1; iy <= ny; ++iy)
        for (int ix = 1; ix <= nx; ++ix) {
            float px = s.dmin_x + ((float)ix - 0.5f) * dx;
            float py = s.dmin_y + ((float)iy - 0.5f) * dy;
            s.func[(size_t)(iy - 1) * nx + (ix - 1)] = maxf(0.0f, filter_evaluate(f, px, py));
        }
    s.marginal_func.assign(ny, 0.0f);
    for (int iy = 0; iy < ny; ++iy)
        for (int ix = 0; ix < nx; ++ix) s.marginal_func[iy] += s.func[(size_t)iy * nx + ix];
    s.marginal_cdf.assign(ny + 1, 0.0f);
    for (int iy = 0; iy < ny; ++iy) s.marginal_cdf[iy + 1] = s.marginal_cdf[iy] + s.marginal_func[iy];
    s.func_integral = s.marginal_cdf[ny] * dx * dy;
    float end = s.marginal_cdf[ny];
    if (end > 0.0f) {
        for (auto& v : s.marginal_cdf) v /= end;
    } else {
        for (int iy = 0; iy <= ny; ++iy) s.marginal_cdf[iy] = (float)iy / (float)ny;
    }
    s.conditional_cdf.assign((size_t)ny * (nx + 1), 0.0f);
    for (int iy = 0; iy < ny; ++iy) {
        float* row = &s.conditional_cdf[(size_t)iy * (nx + 1)];
        row[0] = 0.0f;
        for (int ix = 0; ix < nx; ++ix) row[ix + 1] = row[ix] + s.func[(size_t)iy * nx + ix];
        float rs = row[nx];
        if (rs > 0.0f) {
            for (int ix = 0; ix <= nx; ++ix) row[ix] /= rs;
        } else {
            for (int ix = 0; ix <= nx; ++ix) row[ix] = (float)ix / (float)nx;
        }
    }
    return s;
}

struct FilterSample {
    float px, py, weight;
};
// 20-step branchless binary search; cdf has n+1 entries; returns 1-based lo (filter.jl:733-747)
inline int32_t find_interval20(const float* cdf, float u, int32_t n) {
    int32_t lo = 1, hi = n + 1;
    for (int k = 0; k < 20; ++k) {
        int32_t mid = (lo + hi) >> 1;
        bool cond = cdf[mid - 1] <= u;
        lo = cond ? mid : lo;
        hi = cond ? hi : mid;
    }
    return lo;
}
inline float sample_tent(float u, float r) {
    if (u < 0.5f) {
        float ur = 2.0f * u;
        return -r + r * std::sqrt(ur);
    }
    float ur = 2.0f * (1.0f - u);
    return r * (1.0f - std::sqrt(ur));
}
inline FilterSample filter_sample(const FilterParams& f, const FilterSampler& s, V2 u) {  // filter.jl:876-953
    if (f.type == HK_FILTER_BOX) return FilterSample{lerpf(-f.rx, f.rx, u.x), lerpf(-f.ry, f.ry, u.y), 1.0f};
    if (f.type == HK_FILTER_TRIANGLE) return FilterSample{sample_tent(u.x, f.rx), sample_tent(u.y, f.ry), 1.0f};
    // filter_sample_tabulated  filter.jl:834-870
    int32_t ny = s.ny, nx = s.nx;
    // marginal on u[2]
    int32_t o = find_interval20(s.marginal_cdf.data(), u.y, ny);
    o = clampi(o, 1, ny);
    float du = u.y - s.marginal_cdf[o - 1];
    float diff = s.marginal_cdf[o] - s.marginal_cdf[o - 1];
    du = diff > 0.0f ? du / diff : 0.0f;
    float pdf_y = s.func_integral > 0.0f ? s.marginal_func[o - 1] / s.func_integral : 0.0f;
    float t = ((float)(o - 1) + du) / (float)ny;
    float py = lerpf(s.dmin_y, s.dmax_y, t);
    int32_t iy = o;
    float row_integral = s.marginal_func[iy - 1];
    const float* row = &s.conditional_cdf[(size_t)(iy - 1) * (nx + 1)];
    int32_t lo = find_interval20(row, u.x, nx);
    int32_t ox = clampi(lo, 1, nx);
    float dux = u.x - row[ox - 1];
    float diffx = row[ox] - row[ox - 1];
    dux = diffx > 0.0f ? dux / diffx : 0.0f;
    float fval = s.func[(size_t)(iy - 1) * nx + (ox - 1)];
    float pdf_x = row_integral > 0.0f ? fval / row_integral : 0.0f;
    float tx = ((float)(ox - 1) + dux) / (float)nx;
    float px = lerpf(s.dmin_x, s.dmax_x, tx);
    float pdf = pdf_x * pdf_y;
    float weight = pdf > 0.0f ? fval / pdf : 0.0f;
    return FilterSample{px, py, weight};
}

// ---- camera -----------------------------------------------------------------------------------
// Transformation applied to a Point3f (homogeneous divide) / Vec3f (no translation).  Raycore's
// source is not available; this is pbrt's Transform::operator() which Raycore mirrors:
// the divide is skipped when w == 1.
inline V3 xform_point(const float* m, V3 p) {
    float x = m[0] * p.x + m[1] * p.y + m[2] * p.z + m[3];
    float y = m[4] * p.x + m[5] * p.y + m[6] * p.z + m[7];
    float z = m[8] * p.x + m[9] * p.y + m[10] * p.z + m[11];
    float w = m[12] * p.x + m[13] * p.y + m[14] * p.z + m[15];
    if (w == 1.0f) return V3(x, y, z);
    float inv = 1.0f / w;
    return V3(x * inv, y * inv, z * inv);
}
inline V3 xform_vector(const float* m, V3 v) {
    return V3(m[0] * v.x + m[1] * v.y + m[2] * v.z, m[4] * v.x + m[5] * v.y + m[6] * v.z, m[8] * v.x + m[9] * v.y + m[10] * v.z);
}
struct CamRay {
    V3 o, d;
    float time;
};
inline CamRay generate_ray(const hk_camera& cam, V2 film, V2 lens, float time_u) {  // perspective.jl:95-128
    V3 p_camera = xform_point(cam.raster_to_camera, V3(film.x, film.y, 0.0f));
    V3 o(0.0f);
    V3 d = normalize(p_camera);
    if (cam.lens_radius > 0) {
        V2 dsk = concentric_sample_disk(lens);
        float plx = cam.lens_radius * dsk.x, ply = cam.lens_radius * dsk.y;
        float t = -cam.focal_distance / d.z;
        V3 p_focus = o + d * t;
        o = V3(plx, ply, 0.0f);
        d = normalize(p_focus - o);
    }
    float time = lerpf(cam.shutter_open, cam.shutter_close, time_u);
    CamRay r;
    r.o = xform_point(cam.camera_to_world, o);
    r.d = normalize(xform_vector(cam.camera_to_world, d));
    r.time = time;
    return r;
}

}  // namespace hko

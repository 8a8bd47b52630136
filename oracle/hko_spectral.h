// hko_spectral.h — CPU ORACLE (test infrastructure): wavelength sampling, RGB->spectrum uplift,
// D65, spectral->RGB.  Follows:
//   sample_wavelengths_visible / pdf      src/spectral/spectral.jl:192-249
//   rgb_to_spectrum (table lookup)        src/spectral/rgb2spec.jl:17-53, 71-167
//   uplift_rgb / _unbounded / _illuminant src/spectral/uplift.jl:255-308, 437-475, 514-566
//   sample_d65                            src/spectral/uplift.jl:437-457 (table :393-429)
//   spectral_to_xyz / xyz_to_linear_srgb  src/spectral/color.jl:364-440, 572-579
#pragma once
#include "hko_core.h"

namespace hko {

inline float visible_wavelengths_pdf(float lambda) {
    if (lambda < 360.0f || lambda > 830.0f) return 0.0f;
    float x = 0.0072f * (lambda - 538.0f);
    float c = std::cosh(x);
    return 0.0039398042f / (c * c);
}
inline float sample_visible_wavelengths(float u) { return 538.0f - 138.888889f * std::atanh(0.85691062f - 1.82750197f * u); }
inline Wavelengths sample_wavelengths_visible(float u) {
    Wavelengths w;
    float us[4];
    us[0] = u;
    float u2 = u + 0.25f;
    us[1] = u2 >= 1.0f ? u2 - 1.0f : u2;
    float u3 = u + 0.5f;
    us[2] = u3 >= 1.0f ? u3 - 1.0f : u3;
    float u4 = u + 0.75f;
    us[3] = u4 >= 1.0f ? u4 - 1.0f : u4;
    for (int i = 0; i < 4; ++i) {
        w.lambda[i] = sample_visible_wavelengths(us[i]);
        w.pdf[i] = visible_wavelengths_pdf(w.lambda[i]);
    }
    return w;
}

struct RGB2SpecTable {
    int32_t res;
    const float* scale;   // [res]
    const float* coeffs;  // [3,res,res,res,3] Julia column-major (maxc fastest)
    // coeffs[maxc, zi, yi, xi, c] with 1-based indices
    float at(int maxc, int zi, int yi, int xi, int c) const {
        size_t r = (size_t)res;
        return coeffs[(size_t)(maxc - 1) + 3 * ((size_t)(zi - 1) + r * ((size_t)(yi - 1) + r * ((size_t)(xi - 1) + r * (size_t)(c - 1))))];
    }
};

struct SigPoly {
    float c0, c1, c2;
};
inline float sigmoid(float x) {
    if (std::isinf(x)) return x > 0 ? 1.0f : 0.0f;
    return 0.5f + x / (2.0f * std::sqrt(1.0f + x * x));
}
inline float eval_poly(const SigPoly& p, float lambda) {
    float x = p.c0 * lambda * lambda + p.c1 * lambda + p.c2;
    return sigmoid(x);
}
// max_value(poly)   rgb2spec.jl:39-53
inline float poly_max_value(const SigPoly& p) {
    float result = maxf(eval_poly(p, 360.0f), eval_poly(p, 830.0f));
    if (p.c0 != 0) {
        float lc = -p.c1 / (2.0f * p.c0);
        if (360.0f <= lc && lc <= 830.0f) result = maxf(result, eval_poly(p, lc));
    }
    return result;
}

// rgb_to_spectrum   rgb2spec.jl:85-167
inline SigPoly rgb_to_spectrum(const RGB2SpecTable& t, float r, float g, float b) {
    r = clampf(r, 0.0f, 1.0f);
    g = clampf(g, 0.0f, 1.0f);
    b = clampf(b, 0.0f, 1.0f);
    if (r == g && g == b) {
        float c2;
        if (r > 0.0f && r < 1.0f)
            c2 = (r - 0.5f) / std::sqrt(r * (1.0f - r));
        else if (r <= 0.0f)
            c2 = -1.0e10f;
        else
            c2 = 1.0e10f;
        return SigPoly{0.0f, 0.0f, c2};
    }
    int32_t maxc = r > g ? (r > b ? 1 : 3) : (g > b ? 2 : 3);
    float z = maxc == 1 ? r : (maxc == 2 ? g : b);
    float x_comp = maxc == 1 ? g : (maxc == 2 ? b : r);
    float y_comp = maxc == 1 ? b : (maxc == 2 ? r : g);
    int32_t res = t.res;
    float x = x_comp * (float)(res - 1) / z;
    float y = y_comp * (float)(res - 1) / z;
    int32_t zi = 1;
    for (int32_t i = 1; i <= res - 1; ++i)
        if (t.scale[i - 1] < z) zi = i;
    zi = zi < res - 1 ? zi : res - 1;
    int32_t xi = u_int32(x) + 1;
    xi = xi < res - 1 ? xi : res - 1;
    int32_t yi = u_int32(y) + 1;
    yi = yi < res - 1 ? yi : res - 1;
    float dx = x - (float)(xi - 1);
    float dy = y - (float)(yi - 1);
    float dz = (z - t.scale[zi - 1]) / (t.scale[zi] - t.scale[zi - 1]);
    float c[3];
    for (int k = 1; k <= 3; ++k) {
        c[k - 1] = (1.0f - dz) * ((1.0f - dy) * ((1.0f - dx) * t.at(maxc, zi, yi, xi, k) + dx * t.at(maxc, zi, yi, xi + 1, k)) +
                                  dy * ((1.0f - dx) * t.at(maxc, zi, yi + 1, xi, k) + dx * t.at(maxc, zi, yi + 1, xi + 1, k))) +
                   dz * ((1.0f - dy) * ((1.0f - dx) * t.at(maxc, zi + 1, yi, xi, k) + dx * t.at(maxc, zi + 1, yi, xi + 1, k)) +
                         dy * ((1.0f - dx) * t.at(maxc, zi + 1, yi + 1, xi, k) + dx * t.at(maxc, zi + 1, yi + 1, xi + 1, k)));
    }
    return SigPoly{c[0], c[1], c[2]};
}

// D65, 300..830 nm step 5 (CIE standard illuminant data, uplift.jl:393-429)
static const float D65_VALUES[107] = {
    0.0341f,  1.6643f,  3.2945f,  11.7652f, 20.236f,  28.6447f, 37.0535f, 38.5011f, 39.9488f, 42.4302f, 44.9117f, 45.775f,
    46.6383f, 49.3637f, 52.0891f, 51.0323f, 49.9755f, 52.3118f, 54.6482f, 68.7015f, 82.7549f, 87.1204f, 91.486f,  92.4589f,
    93.4318f, 90.057f,  86.6823f, 95.7736f, 104.865f, 110.936f, 117.008f, 117.41f,  117.812f, 116.336f, 114.861f, 115.392f,
    115.923f, 112.367f, 108.811f, 109.082f, 109.354f, 108.578f, 107.802f, 106.296f, 104.79f,  106.239f, 107.689f, 106.047f,
    104.405f, 104.225f, 104.046f, 102.023f, 100.0f,   98.1671f, 96.3342f, 96.0611f, 95.788f,  92.2368f, 88.6856f, 89.3459f,
    90.0062f, 89.8026f, 89.5991f, 88.6489f, 87.6987f, 85.4936f, 83.2886f, 83.4939f, 83.6992f, 81.863f,  80.0268f, 80.1207f,
    80.2146f, 81.2462f, 82.2778f, 80.281f,  78.2842f, 74.0027f, 69.7213f, 70.6652f, 71.6091f, 72.979f,  74.349f,  67.9765f,
    61.604f,  65.7448f, 69.8856f, 72.4863f, 75.087f,  69.3398f, 63.5927f, 55.0054f, 46.4182f, 56.6118f, 66.8054f, 65.0941f,
    63.3828f, 63.8434f, 64.304f,  61.8779f, 59.4519f, 55.7054f, 51.959f,  54.6998f, 57.4406f, 58.8765f, 60.3125f};

inline float sample_d65(float lambda) {  // uplift.jl:437-457
    if (lambda <= 300.0f) return D65_VALUES[0];
    if (lambda >= 830.0f) return D65_VALUES[106];
    float t = (lambda - 300.0f) / 5.0f;
    int32_t idx = floor_int32(t) + 1;
    idx = clampi(idx, 1, 106);
    float frac = t - (float)floor_int32(t);
    float v0 = D65_VALUES[idx - 1], v1 = D65_VALUES[idx];
    return v0 * (1.0f - frac) + v1 * frac;
}

// uplift_rgb (bounded; clamps rgb to [0,1] inside rgb_to_spectrum)   uplift.jl:255-266, 348-352
inline Spec uplift_rgb(const RGB2SpecTable& t, const RGBA& rgb, const Wavelengths& w) {
    SigPoly p = rgb_to_spectrum(t, rgb.c[0], rgb.c[1], rgb.c[2]);
    return Spec(eval_poly(p, w.lambda[0]), eval_poly(p, w.lambda[1]), eval_poly(p, w.lambda[2]), eval_poly(p, w.lambda[3]));
}
// uplift_rgb_unbounded   uplift.jl:286-308 (quirk Q5)
inline Spec uplift_rgb_unbounded(const RGB2SpecTable& t, const RGBA& rgb, const Wavelengths& w) {
    float r = rgb.c[0], g = rgb.c[1], b = rgb.c[2];
    float m = maxf(maxf(r, g), b);
    if (m <= 0.0f) return Spec(0.0f);
    SigPoly p = rgb_to_spectrum(t, r / m, g / m, b / m);
    float max_poly = poly_max_value(p);
    float scale = m / max_poly;
    return Spec(scale * eval_poly(p, w.lambda[0]), scale * eval_poly(p, w.lambda[1]), scale * eval_poly(p, w.lambda[2]),
                scale * eval_poly(p, w.lambda[3]));
}
// rgb_to_spectral_sigmoid_illuminant   uplift.jl:514-538
inline Spec uplift_rgb_illuminant(const RGB2SpecTable& t, const RGBA& rgb, const Wavelengths& w) {
    float r = rgb.c[0], g = rgb.c[1], b = rgb.c[2];
    float m = maxf(maxf(r, g), b);
    if (m <= 0.0f) return Spec(0.0f);
    float scale = 2.0f * m;
    SigPoly p = rgb_to_spectrum(t, r / scale, g / scale, b / scale);
    Spec out;
    for (int i = 0; i < 4; ++i) out.v[i] = scale * eval_poly(p, w.lambda[i]) * sample_d65(w.lambda[i]);
    return out;
}
// Sample(::RGBIlluminantSpectrum, lambda)   uplift.jl:487-497
inline Spec sample_illuminant(const SigPoly& p, float scale, const Wavelengths& w) {
    Spec out;
    for (int i = 0; i < 4; ++i) out.v[i] = scale * eval_poly(p, w.lambda[i]) * sample_d65(w.lambda[i]);
    return out;
}

struct CIETable {
    const float *x, *y, *z;  // 471 each
};
inline float sample_cie(const float* tab, float lambda) {  // color.jl:364-395
    int32_t offset = round_int32(lambda) - 360;
    if (offset < 0 || offset >= 471) return 0.0f;
    return tab[offset];
}
// spectral_to_xyz  color.jl:418-434 (no division by CIE_Y_INTEGRAL, quirk Q14)
inline V3 spectral_to_xyz(const CIETable& c, const Spec& L, const Wavelengths& w) {
    V3 s(0.0f);
    for (int i = 0; i < 4; ++i) {
        float pdf = w.pdf[i];
        if (pdf != 0.0f) {
            V3 cmf(sample_cie(c.x, w.lambda[i]), sample_cie(c.y, w.lambda[i]), sample_cie(c.z, w.lambda[i]));
            s = s + (cmf * L.v[i]) / pdf;
        }
    }
    return s * 0.25f;
}
inline V3 xyz_to_linear_srgb(V3 xyz) {  // color.jl:572-579
    float X = xyz.x, Y = xyz.y, Z = xyz.z;
    return V3(3.2404542f * X - 1.5371385f * Y - 0.4985314f * Z, -0.9692660f * X + 1.8760108f * Y + 0.0415560f * Z,
              0.0556434f * X - 0.2040259f * Y + 1.0572252f * Z);
}

}  // namespace hko

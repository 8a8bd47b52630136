// hko_core.h — CPU ORACLE (test infrastructure, never shipped or linked into the product).
//
// Plain C++17 restatement of the small numeric types the reference's VolPath path is written in.
// Arithmetic is strict binary32, evaluated in the order the Julia source evaluates it; compile with
// -ffp-contract=off (no FMA contraction: Julia does not contract either).
//
//   Vec3f / Point3f  (GeometryBasics static vectors): dot = (a1*b1 + a2*b2) + a3*b3,
//                    normalize(v) = (1/norm(v)) * v   (StaticArrays: inv(norm(a))*a)
//   SampledSpectrum{4} / SpectralRadiance   src/spectral/spectral.jl:10-111
//   SampledWavelengths{4}                   src/spectral/spectral.jl:121-126
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>

namespace hko {

struct V3 {
    float x, y, z;
    V3() : x(0), y(0), z(0) {}
    V3(float a, float b, float c) : x(a), y(b), z(c) {}
    explicit V3(float a) : x(a), y(a), z(a) {}
    float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
};
inline V3 operator+(V3 a, V3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline V3 operator-(V3 a, V3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline V3 operator-(V3 a) { return V3(-a.x, -a.y, -a.z); }
inline V3 operator*(V3 a, float s) { return V3(a.x * s, a.y * s, a.z * s); }
inline V3 operator*(float s, V3 a) { return V3(s * a.x, s * a.y, s * a.z); }
inline V3 operator/(V3 a, float s) { return V3(a.x / s, a.y / s, a.z / s); }
inline float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 cross(V3 a, V3 b) { return V3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
inline float norm(V3 a) { return std::sqrt(dot(a, a)); }
inline V3 normalize(V3 a) { return (1.0f / norm(a)) * a; }
inline bool operator==(V3 a, V3 b) { return a.x == b.x && a.y == b.y && a.z == b.z; }
inline bool operator!=(V3 a, V3 b) { return !(a == b); }

struct V2 {
    float x, y;
    V2() : x(0), y(0) {}
    V2(float a, float b) : x(a), y(b) {}
};

// Julia's clamp / min / max on Float32 (no NaN subtleties needed on this path)
inline float clampf(float v, float lo, float hi) { return v < lo ? lo : (v > hi ? hi : v); }
inline float maxf(float a, float b) { return a > b ? a : b; }  // Julia max(a,b): NaN-propagating; inputs here are finite
inline float minf(float a, float b) { return a < b ? a : b; }
inline int32_t clampi(int32_t v, int32_t lo, int32_t hi) { return v < lo ? lo : (v > hi ? hi : v); }
// lerp(v1, v2, t) = (1 - t) * v1 + t * v2          src/spectrum.jl:33
inline float lerpf(float v1, float v2, float t) { return (1.0f - t) * v1 + t * v2; }
// Base.unsafe_trunc(Int32, floor(x)) / round(x) (ties-to-even)   src/materials/bsdf.jl:126-135
inline int32_t floor_int32(float x) { return (int32_t)std::floor(x); }
inline int32_t round_int32(float x) { return (int32_t)std::nearbyint(x); }
inline int32_t u_int32(float x) { return (int32_t)x; }

inline uint32_t f2u(float f) {
    uint32_t u;
    std::memcpy(&u, &f, 4);
    return u;
}

static const float PI_F = 3.14159265358979323846f;  // Float32(pi)
static const float INF_F = std::numeric_limits<float>::infinity();

// ---- SampledSpectrum{4} -------------------------------------------------------------------
struct Spec {
    float v[4];
    Spec() : v{0, 0, 0, 0} {}
    explicit Spec(float a) : v{a, a, a, a} {}
    Spec(float a, float b, float c, float d) : v{a, b, c, d} {}
    float operator[](int i) const { return v[i]; }
};
#define HKO_SPEC_OP(op)                                                                      \
    inline Spec operator op(const Spec& a, const Spec& b) {                                  \
        return Spec(a.v[0] op b.v[0], a.v[1] op b.v[1], a.v[2] op b.v[2], a.v[3] op b.v[3]); \
    }
HKO_SPEC_OP(+)
HKO_SPEC_OP(-)
HKO_SPEC_OP(*)
HKO_SPEC_OP(/)
#undef HKO_SPEC_OP
inline Spec operator*(const Spec& a, float s) { return Spec(a.v[0] * s, a.v[1] * s, a.v[2] * s, a.v[3] * s); }
inline Spec operator*(float s, const Spec& a) { return a * s; }  // spectral.jl:48: s*a = a*s
inline Spec operator/(const Spec& a, float s) { return Spec(a.v[0] / s, a.v[1] / s, a.v[2] / s, a.v[3] / s); }
inline Spec operator-(const Spec& a) { return Spec(-a.v[0], -a.v[1], -a.v[2], -a.v[3]); }
inline Spec exp(const Spec& a) { return Spec(std::exp(a.v[0]), std::exp(a.v[1]), std::exp(a.v[2]), std::exp(a.v[3])); }
inline Spec sqrt(const Spec& a) { return Spec(std::sqrt(a.v[0]), std::sqrt(a.v[1]), std::sqrt(a.v[2]), std::sqrt(a.v[3])); }
// average = sum(data)/N ; Julia sum over a 4-tuple is ((a+b)+c)+d      spectral.jl:67-69
inline float average(const Spec& s) { return (((s.v[0] + s.v[1]) + s.v[2]) + s.v[3]) / 4; }
inline float max_component(const Spec& s) { return maxf(maxf(maxf(s.v[0], s.v[1]), s.v[2]), s.v[3]); }
inline bool is_black(const Spec& s) { return s.v[0] == 0.0f && s.v[1] == 0.0f && s.v[2] == 0.0f && s.v[3] == 0.0f; }
inline Spec safe_div(const Spec& a, const Spec& b) {
    Spec r;
    for (int i = 0; i < 4; ++i) r.v[i] = b.v[i] != 0.0f ? a.v[i] / b.v[i] : 0.0f;
    return r;
}

struct Wavelengths {
    float lambda[4];
    float pdf[4];
};

// RGBSpectrum: r,g,b,alpha     src/spectrum.jl:38-43
struct RGBA {
    float c[4];
    RGBA() : c{0, 0, 0, 1} {}
    RGBA(float r, float g, float b, float a = 1.0f) : c{r, g, b, a} {}
};
inline RGBA operator*(const RGBA& a, float s) { return RGBA(a.c[0] * s, a.c[1] * s, a.c[2] * s, a.c[3] * s); }
// clamp(::RGBSpectrum) defaults to [0, Inf)   src/spectrum.jl:61-70 (quirk Q25)
inline RGBA clamp_rgb(const RGBA& a) {
    return RGBA(clampf(a.c[0], 0.0f, INF_F), clampf(a.c[1], 0.0f, INF_F), clampf(a.c[2], 0.0f, INF_F), clampf(a.c[3], 0.0f, INF_F));
}
// luminance(::RGBSpectrum)     src/lights/light-sampler.jl:448-450
inline float luminance(const RGBA& s) { return 0.212671f * s.c[0] + 0.715160f * s.c[1] + 0.072169f * s.c[2]; }

// ------------------------------------------------------------------------------------------------
// sin / cos of a binary32 argument the way Julia's Base computes them (base/special/trig.jl, a port of FreeBSD msun's k_sinf /
// k_cosf / e_rem_pio2f): argument reduction and the polynomial kernels in binary64, ONE rounding to binary32 at the end.  The
// reference calls Base.sin / Base.cos on Float32: this is that arithmetic (CPU ORACLE, test infrastructure).
// ------------------------------------------------------------------------------------------------
inline float jl_sin_kernel(double y) {
    const double S1 = -0.16666666641626524, S2 = 0.008333329385889463, S3 = -0.00019839334836096632, S4 = 2.718311493989822e-6;
    double z = y * y, w = z * z;
    double r = S3 + z * S4, s = z * y;
    return (float)((y + s * (S1 + z * S2)) + s * w * r);
}
inline float jl_cos_kernel(double y) {
    const double C0 = -0.499999997251031, C1 = 0.04166662332373906, C2 = -0.001388676377460993, C3 = 2.439044879627741e-5;
    double z = y * y, w = z * z;
    double r = C2 + z * C3;
    return (float)(((1.0 + z * C0) + w * C1) + (w * z) * r);
}
// rem_pio2_kernel(x::Float32): n and the reduced argument (binary64) with x = n * pi/2 + y, |x| < 2^28 * pi/2
inline int jl_rem_pio2(float x, double& y) {
    const double PI = 3.141592653589793;
    const double xd = (double)x, ax = std::fabs(xd);
    if (ax <= PI * 5 / 4) {
        if (ax <= PI * 3 / 4) {
            y = x > 0 ? xd - PI / 2 : xd + PI / 2;
            return x > 0 ? 1 : -1;
        }
        y = x > 0 ? xd - PI : xd + PI;
        return x > 0 ? 2 : -2;
    }
    if (ax <= PI * 9 / 4) {
        if (ax <= PI * 7 / 4) {
            y = x > 0 ? xd - 3 * (PI / 2) : xd + 3 * (PI / 2);
            return x > 0 ? 3 : -3;
        }
        y = x > 0 ? xd - 2 * PI : xd + 2 * PI;
        return x > 0 ? 4 : -4;
    }
    const double fn = std::rint(xd * 6.36619772367581382433e-01);   // Cody-Waite with a 33 + 53 bit pi/2
    const double r = xd - fn * 1.57079631090164184570e+00, w = fn * 1.58932547735281966916e-08;
    y = r - w;
    return (int)fn;
}
inline void jl_sincos(float x, float& s, float& c) {
    const float ax = std::fabs(x);
    if (ax < 0.7853982f) {   // Float32(pi)/4: no reduction
        s = ax < 0.00034526698f ? x : jl_sin_kernel((double)x);      // sqrt(eps(Float32))
        c = ax < 0.00024414062f ? 1.0f : jl_cos_kernel((double)x);   // sqrt(eps(Float32)/2)
        return;
    }
    if (!(ax < 2.1e8f)) {   // beyond the medium range (never reached by the sampling code): libm
        s = std::sin(x);
        c = std::cos(x);
        return;
    }
    double y;
    const int n = jl_rem_pio2(x, y) & 3;
    const float sk = jl_sin_kernel(y), ck = jl_cos_kernel(y);
    s = n == 0 ? sk : (n == 1 ? ck : (n == 2 ? -sk : -ck));
    c = n == 0 ? ck : (n == 1 ? -sk : (n == 2 ? -ck : sk));
}

}  // namespace hko

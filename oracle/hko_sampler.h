// hko_sampler.h — CPU ORACLE (test infrastructure): hashes, RNGs, ZSobol sampler, sampling prims.
// Follows, line by line:
//   murmur_hash_64a / mix_bits / pbrt_hash / PCG32   src/materials/spectral-eval.jl:575-815
//   LCG (delta tracking)                             src/integrators/volpath/delta-tracking.jl:18-58
//   ZSobol                                           src/sampler/sobol.jl:17-323, 411-447
//   concentric disk / cosine hemisphere              src/sampler/sampling.jl:1-30
#pragma once
#include "hko_core.h"

namespace hko {

// ---- MurmurHash64A  (spectral-eval.jl:575-633) ----------------------------------------------
inline uint64_t murmur_hash_64a(const uint8_t* data, int n, uint64_t seed = 0) {
    const uint64_t m = 0xc6a4a7935bd1e995ull;
    const int r = 47;
    uint64_t h = seed ^ ((uint64_t)n * m);
    int n_chunks = n / 8;
    for (int i = 0; i < n_chunks; ++i) {
        uint64_t k = 0;
        for (int b = 0; b < 8; ++b) k |= (uint64_t)data[8 * i + b] << (8 * b);
        k *= m;
        k ^= k >> r;
        k *= m;
        h ^= k;
        h *= m;
    }
    int remaining = n & 7;
    const uint8_t* tail = data + 8 * n_chunks;
    if (remaining >= 7) h ^= (uint64_t)tail[6] << 48;
    if (remaining >= 6) h ^= (uint64_t)tail[5] << 40;
    if (remaining >= 5) h ^= (uint64_t)tail[4] << 32;
    if (remaining >= 4) h ^= (uint64_t)tail[3] << 24;
    if (remaining >= 3) h ^= (uint64_t)tail[2] << 16;
    if (remaining >= 2) h ^= (uint64_t)tail[1] << 8;
    if (remaining >= 1) {
        h ^= (uint64_t)tail[0];
        h *= m;
    }
    h ^= h >> r;
    h *= m;
    h ^= h >> r;
    return h;
}

// mix_bits (spectral-eval.jl:641-648)
inline uint64_t mix_bits(uint64_t v) {
    v ^= v >> 31;
    v *= 0x7fb5d329728ea185ull;
    v ^= v >> 27;
    v *= 0x81dadef4bc2dd44dull;
    v ^= v >> 33;
    return v;
}

inline void put_f32(uint8_t* b, float f) {
    uint32_t u = f2u(f);
    b[0] = u & 0xff;
    b[1] = (u >> 8) & 0xff;
    b[2] = (u >> 16) & 0xff;
    b[3] = (u >> 24) & 0xff;
}
inline void put_u64(uint8_t* b, uint64_t v) {
    for (int i = 0; i < 8; ++i) b[i] = (v >> (8 * i)) & 0xff;
}
// pbrt_hash overloads (spectral-eval.jl:690-741)
inline uint64_t pbrt_hash(float v) {
    uint8_t b[4];
    put_f32(b, v);
    return murmur_hash_64a(b, 4);
}
inline uint64_t pbrt_hash(uint64_t v) {
    uint8_t b[8];
    put_u64(b, v);
    return murmur_hash_64a(b, 8);
}
inline uint64_t pbrt_hash(V3 v) {
    uint8_t b[12];
    put_f32(b, v.x);
    put_f32(b + 4, v.y);
    put_f32(b + 8, v.z);
    return murmur_hash_64a(b, 12);
}
inline uint64_t pbrt_hash(uint64_t seed, V3 v) {
    uint8_t b[20];
    put_u64(b, seed);
    put_f32(b + 8, v.x);
    put_f32(b + 12, v.y);
    put_f32(b + 16, v.z);
    return murmur_hash_64a(b, 20);
}
inline uint64_t pbrt_hash(uint64_t a, float f) {
    uint8_t b[12];
    put_u64(b, a);
    put_f32(b + 8, f);
    return murmur_hash_64a(b, 12);
}
inline uint64_t pbrt_hash(float a, V2 p) {
    uint8_t b[12];
    put_f32(b, a);
    put_f32(b + 4, p.x);
    put_f32(b + 8, p.y);
    return murmur_hash_64a(b, 12);
}

// ---- PCG32 (spectral-eval.jl:745-815) --------------------------------------------------------
struct PCG32 {
    uint64_t state, inc;
};
static const uint64_t PCG32_MULT = 0x5851f42d4c957f2dull;
inline PCG32 pcg32_init(uint64_t seq_index, uint64_t seed) {
    PCG32 r;
    r.inc = (seq_index << 1) | 1ull;
    uint64_t state = 0;
    state = state * PCG32_MULT + r.inc;
    state += seed;
    state = state * PCG32_MULT + r.inc;
    r.state = state;
    return r;
}
inline PCG32 pcg32_init(uint64_t seq_index) { return pcg32_init(seq_index, mix_bits(seq_index)); }
inline uint32_t pcg32_uniform_u32(PCG32& rng) {
    uint64_t oldstate = rng.state;
    rng.state = oldstate * PCG32_MULT + rng.inc;
    uint64_t xorshifted = ((oldstate >> 18) ^ oldstate) >> 27;
    uint64_t rot = oldstate >> 59;
    uint32_t x32 = (uint32_t)(xorshifted & 0xFFFFFFFFull);
    uint32_t rot32 = (uint32_t)(rot & 0x1F);
    return (x32 >> rot32) | (x32 << ((32 - rot32) & 31));
}
inline float pcg32_uniform_f32(PCG32& rng) {
    uint32_t u = pcg32_uniform_u32(rng);
    // min(1 - eps(Float32), Float32(u32) * 2.3283064f-10)
    float f = (float)u * 2.3283064e-10f;
    const float lim = 1.0f - 1.1920929e-7f;
    return f < lim ? f : lim;
}

// ---- ZSobol (sampler/sobol.jl) ---------------------------------------------------------------
static const float FLOAT32_SCALE = 2.3283064365386963e-10f;
static const float SOBOL_ONE_MINUS_EPS = 1.0f - 1.1920929e-7f;  // Float32(1.0) - eps(Float32)

inline uint64_t zsobol_hash(int32_t dimension, uint32_t seed) {  // sobol.jl:17-31
    uint8_t b[8];
    uint32_t d = (uint32_t)dimension;
    for (int i = 0; i < 4; ++i) b[i] = (d >> (8 * i)) & 0xff;
    for (int i = 0; i < 4; ++i) b[4 + i] = (seed >> (8 * i)) & 0xff;
    return murmur_hash_64a(b, 8, 0);
}
inline uint64_t left_shift2(uint64_t x) {  // sobol.jl:42-50
    x &= 0xffffffffull;
    x = (x ^ (x << 16)) & 0x0000ffff0000ffffull;
    x = (x ^ (x << 8)) & 0x00ff00ff00ff00ffull;
    x = (x ^ (x << 4)) & 0x0f0f0f0f0f0f0f0full;
    x = (x ^ (x << 2)) & 0x3333333333333333ull;
    x = (x ^ (x << 1)) & 0x5555555555555555ull;
    return x;
}
inline uint64_t encode_morton2(uint32_t x, uint32_t y) { return (left_shift2(y) << 1) | left_shift2(x); }
inline uint32_t bitreverse32(uint32_t v) {
    v = ((v >> 1) & 0x55555555u) | ((v & 0x55555555u) << 1);
    v = ((v >> 2) & 0x33333333u) | ((v & 0x33333333u) << 2);
    v = ((v >> 4) & 0x0f0f0f0fu) | ((v & 0x0f0f0f0fu) << 4);
    v = ((v >> 8) & 0x00ff00ffu) | ((v & 0x00ff00ffu) << 8);
    return (v >> 16) | (v << 16);
}
inline uint32_t fast_owen_scramble(uint32_t v, uint32_t seed) {  // sobol.jl:72-80
    v = bitreverse32(v);
    v ^= v * 0x3d20adeau;
    v += seed;
    v *= (seed >> 16) | 1u;
    v ^= v * 0x05526c56u;
    v ^= v * 0x53a22864u;
    return bitreverse32(v);
}
static const int SOBOL_MATRIX_SIZE = 52;
inline float sobol_sample(int64_t a, int32_t dimension, uint32_t scramble_seed, const uint32_t* matrices) {  // sobol.jl:108-127
    uint32_t v = 0;
    int base = dimension * SOBOL_MATRIX_SIZE;
    for (int bit0 = 0; bit0 < SOBOL_MATRIX_SIZE; ++bit0) {
        uint32_t bit_val = (uint32_t)((a >> bit0) & 1);
        uint32_t mask = bit_val * 0xffffffffu;
        v ^= matrices[base + bit0] & mask;
    }
    v = fast_owen_scramble(v, scramble_seed);
    float f = (float)v * FLOAT32_SCALE;
    return f < SOBOL_ONE_MINUS_EPS ? f : SOBOL_ONE_MINUS_EPS;
}
static const uint8_t PERMUTATIONS_4WAY[24][4] = {  // sobol.jl:155-180
    {0, 1, 2, 3}, {0, 1, 3, 2}, {0, 2, 1, 3}, {0, 2, 3, 1}, {0, 3, 2, 1}, {0, 3, 1, 2}, {1, 0, 2, 3}, {1, 0, 3, 2},
    {1, 2, 0, 3}, {1, 2, 3, 0}, {1, 3, 2, 0}, {1, 3, 0, 2}, {2, 1, 0, 3}, {2, 1, 3, 0}, {2, 0, 1, 3}, {2, 0, 3, 1},
    {2, 3, 0, 1}, {2, 3, 1, 0}, {3, 1, 2, 0}, {3, 1, 0, 2}, {3, 2, 1, 0}, {3, 2, 0, 1}, {3, 0, 2, 1}, {3, 0, 1, 2}};

inline uint64_t zsobol_get_sample_index(uint64_t morton_index, int32_t dimension, int32_t log2_spp, int32_t n_base4_digits) {  // sobol.jl:211-258
    uint64_t sample_index = 0;
    int32_t pow2_flag = log2_spp & 1;
    int32_t last_digit = pow2_flag;
    int32_t pow2_adjust = pow2_flag;
    for (int32_t iter0 = 0; iter0 < 32; ++iter0) {
        int32_t i = n_base4_digits - 1 - iter0;
        int32_t raw_shift = 2 * i - pow2_adjust;
        int32_t digit_shift = raw_shift > 0 ? raw_shift : 0;
        int32_t digit = (int32_t)((morton_index >> digit_shift) & 3ull);
        // Julia: x >> n with n >= 64 gives 0
        int32_t hs = digit_shift + 2;
        uint64_t higher_digits = hs >= 64 ? 0ull : (morton_index >> hs);
        uint64_t hash_val = mix_bits(higher_digits ^ (0x55555555ull * (uint64_t)(int64_t)dimension));
        int32_t p = (int32_t)((hash_val >> 24) % 24ull);
        uint64_t permuted_digit = PERMUTATIONS_4WAY[p][digit];
        if (i >= last_digit) sample_index |= permuted_digit << digit_shift;
    }
    uint64_t digit = morton_index & 1ull;
    uint64_t xor_bit = mix_bits((morton_index >> 1) ^ (0x55555555ull * (uint64_t)(int64_t)dimension)) & 1ull;
    if (pow2_flag) sample_index |= (digit ^ xor_bit);
    return sample_index;
}

struct SobolRNG {  // sobol.jl:349-379
    const uint32_t* matrices;
    int32_t log2_spp, n_base4_digits;
    uint32_t seed;
    int32_t width;
};
inline int32_t ceil_log2(int64_t v) {  // Int32(ceil(Int, log2(max(1, v))))
    int32_t l = 0;
    while (((int64_t)1 << l) < v) ++l;
    return l;
}
inline SobolRNG make_sobol_rng(const uint32_t* matrices, uint32_t seed, int width, int height, int spp) {  // sobol.jl:317-323
    SobolRNG r;
    r.matrices = matrices;
    r.log2_spp = ceil_log2(spp < 1 ? 1 : spp);
    int32_t res_log2 = ceil_log2(width > height ? width : height);
    int32_t log4_spp = (r.log2_spp + 1) / 2;
    r.n_base4_digits = res_log2 + log4_spp;
    r.seed = seed;
    r.width = width;
    return r;
}
inline float sample_1d(const SobolRNG& r, int32_t px, int32_t py, int32_t sample_idx, int32_t dim) {  // sobol.jl:269-282
    uint64_t morton = (encode_morton2((uint32_t)px, (uint32_t)py) << r.log2_spp) | (uint64_t)(int64_t)sample_idx;
    uint64_t idx = zsobol_get_sample_index(morton, dim, r.log2_spp, r.n_base4_digits);
    uint32_t h = (uint32_t)zsobol_hash(dim + 1, r.seed);
    return sobol_sample((int64_t)idx, 0, h, r.matrices);
}
inline V2 sample_2d(const SobolRNG& r, int32_t px, int32_t py, int32_t sample_idx, int32_t dim) {  // sobol.jl:290-309
    uint64_t morton = (encode_morton2((uint32_t)px, (uint32_t)py) << r.log2_spp) | (uint64_t)(int64_t)sample_idx;
    uint64_t idx = zsobol_get_sample_index(morton, dim, r.log2_spp, r.n_base4_digits);
    uint64_t bits = zsobol_hash(dim + 2, r.seed);
    uint32_t h1 = (uint32_t)bits, h2 = (uint32_t)(bits >> 32);
    return V2(sobol_sample((int64_t)idx, 0, h1, r.matrices), sobol_sample((int64_t)idx, 1, h2, r.matrices));
}

// ---- sampling primitives (sampler/sampling.jl:1-30) -----------------------------------------
inline V2 concentric_sample_disk(V2 u) {
    float ox = 2.0f * u.x - 1.0f, oy = 2.0f * u.y - 1.0f;
    float ax = std::fabs(ox), ay = std::fabs(oy);
    float sx = ox + 1.0e-10f, sy = oy + 1.0e-10f;
    bool xl = ax > ay;
    float r = xl ? ox : oy;
    float theta = xl ? ((oy / sx) * PI_F) / 4.0f : PI_F / 2.0f - ((ox / sy) * PI_F) / 4.0f;
    float st, ct;
    jl_sincos(theta, st, ct);
    return V2(r * ct, r * st);
}
inline V3 cosine_sample_hemisphere(V2 u) {
    V2 d = concentric_sample_disk(u);
    float z = std::sqrt(maxf(0.0f, 1.0f - d.x * d.x - d.y * d.y));
    return V3(d.x, d.y, z);
}

}  // namespace hko

// hk_device.h — device functions of the MI355X VolPath path (gfx950, wave64).
//
// Everything here is strict binary32 in the evaluation order of the Julia source it replaces (the
// translation unit is built with -ffp-contract=off; the only fused operations are the explicit
// fmaf() calls in the BVH slab test, which is result-neutral culling).  Reference lines are cited
// per function; behaviour quirks are the ones SURVEY.md §8-Q lists.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "hk_types.h"

#include "hk_nanovdb.h"

#define HKD __device__ __forceinline__

namespace hkd {

static constexpr float PI_F = 3.14159265358979323846f;
static constexpr float INF_F = __builtin_huge_valf();

struct v3 {
    float x, y, z;
};
struct v2 {
    float x, y;
};
HKD v3 mk3(float x, float y, float z) { return v3{x, y, z}; }
HKD v2 mk2(float x, float y) { return v2{x, y}; }
HKD v3 operator+(v3 a, v3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
HKD v3 operator-(v3 a, v3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
HKD v3 operator-(v3 a) { return mk3(-a.x, -a.y, -a.z); }
HKD v3 operator*(v3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
HKD v3 operator*(float s, v3 a) { return mk3(s * a.x, s * a.y, s * a.z); }
HKD v3 operator/(v3 a, float s) { return mk3(a.x / s, a.y / s, a.z / s); }
HKD float dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
HKD v3 cross(v3 a, v3 b) { return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
HKD float norm(v3 a) { return sqrtf(dot(a, a)); }
HKD v3 normalize(v3 a) { return (1.0f / norm(a)) * a; }  // StaticArrays: inv(norm(a)) * a
HKD bool is_zero(v3 a) { return a.x == 0.0f && a.y == 0.0f && a.z == 0.0f; }
HKD float comp(v3 a, int k) { return k == 0 ? a.x : (k == 1 ? a.y : a.z); }

HKD float clampf(float v, float lo, float hi) { return v < lo ? lo : (v > hi ? hi : v); }
HKD float maxf(float a, float b) { return a > b ? a : b; }
HKD float minf(float a, float b) { return a < b ? a : b; }
HKD int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
HKD float lerpf(float v1, float v2, float t) { return (1.0f - t) * v1 + t * v2; }  // spectrum.jl:33

// SampledSpectrum{4}  (spectral/spectral.jl:10-111)
struct S4 {
    float x, y, z, w;
};
HKD S4 s4(float a) { return S4{a, a, a, a}; }
HKD S4 s4(float a, float b, float c, float d) { return S4{a, b, c, d}; }
HKD S4 ld4(const float4* p) {
    float4 v = *p;
    return S4{v.x, v.y, v.z, v.w};
}
HKD void st4(float4* p, S4 s) { *p = make_float4(s.x, s.y, s.z, s.w); }
HKD S4 operator+(S4 a, S4 b) { return S4{a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; }
HKD S4 operator-(S4 a, S4 b) { return S4{a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w}; }
HKD S4 operator*(S4 a, S4 b) { return S4{a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w}; }
HKD S4 operator/(S4 a, S4 b) { return S4{a.x / b.x, a.y / b.y, a.z / b.z, a.w / b.w}; }
HKD S4 operator*(S4 a, float s) { return S4{a.x * s, a.y * s, a.z * s, a.w * s}; }
HKD S4 operator*(float s, S4 a) { return a * s; }
HKD S4 operator/(S4 a, float s) { return S4{a.x / s, a.y / s, a.z / s, a.w / s}; }
HKD float average(S4 s) { return (((s.x + s.y) + s.z) + s.w) / 4.0f; }
HKD float max_component(S4 s) { return maxf(maxf(maxf(s.x, s.y), s.z), s.w); }
HKD bool is_black(S4 s) { return s.x == 0.0f && s.y == 0.0f && s.z == 0.0f && s.w == 0.0f; }
HKD float at(S4 s, int i) { return i == 0 ? s.x : (i == 1 ? s.y : (i == 2 ? s.z : s.w)); }

// ------------------------------------------------------------------------------------------------
// hashes / RNG  (materials/spectral-eval.jl:575-815)
// ------------------------------------------------------------------------------------------------
HKD uint64_t mix_bits(uint64_t v) {
    v ^= v >> 31;
    v *= 0x7fb5d329728ea185ull;
    v ^= v >> 27;
    v *= 0x81dadef4bc2dd44dull;
    v ^= v >> 33;
    return v;
}
// MurmurHash64A over little-endian 32-bit words (all call sites hash 4-byte multiples)
template <int NWORDS>
HKD uint64_t murmur64a_words(const uint32_t* w) {
    const uint64_t m = 0xc6a4a7935bd1e995ull;
    const int r = 47;
    uint64_t h = 0ull ^ ((uint64_t)(NWORDS * 4) * m);
#pragma unroll
    for (int i = 0; i < NWORDS / 2; ++i) {
        uint64_t k = (uint64_t)w[2 * i] | ((uint64_t)w[2 * i + 1] << 32);
        k *= m;
        k ^= k >> r;
        k *= m;
        h ^= k;
        h *= m;
    }
    if (NWORDS & 1) {  // 4 trailing bytes: the switch fall-through xors bytes 3..0, multiplies once
        h ^= (uint64_t)w[NWORDS - 1];
        h *= m;
    }
    h ^= h >> r;
    h *= m;
    h ^= h >> r;
    return h;
}
HKD uint64_t pbrt_hash(v3 v) {
    uint32_t w[3] = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z)};
    return murmur64a_words<3>(w);
}
HKD uint64_t pbrt_hash(uint64_t seed, v3 v) {
    uint32_t w[5] = {(uint32_t)seed, (uint32_t)(seed >> 32), __float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z)};
    return murmur64a_words<5>(w);
}
HKD uint64_t pbrt_hash(float a, v2 p) {
    uint32_t w[3] = {__float_as_uint(a), __float_as_uint(p.x), __float_as_uint(p.y)};
    return murmur64a_words<3>(w);
}
struct PCG32 {
    uint64_t state, inc;
};
HKD PCG32 pcg32_init(uint64_t seq, uint64_t seed) {
    const uint64_t M = 0x5851f42d4c957f2dull;
    PCG32 r;
    r.inc = (seq << 1) | 1ull;
    uint64_t s = 0;
    s = s * M + r.inc;
    s += seed;
    s = s * M + r.inc;
    r.state = s;
    return r;
}
HKD uint32_t pcg32_u32(PCG32& g) {
    uint64_t old = g.state;
    g.state = old * 0x5851f42d4c957f2dull + g.inc;
    uint32_t xs = (uint32_t)((((old >> 18) ^ old) >> 27) & 0xFFFFFFFFull);
    uint32_t rot = (uint32_t)(old >> 59) & 31u;
    return (xs >> rot) | (xs << ((32 - rot) & 31));
}
HKD float pcg32_f32(PCG32& g) {
    float f = (float)pcg32_u32(g) * 2.3283064e-10f;
    const float lim = 1.0f - 1.1920929e-7f;
    return f < lim ? f : lim;
}

// ------------------------------------------------------------------------------------------------
// ZSobol  (sampler/sobol.jl:17-323).  Only Sobol dimensions 0 and 1 are ever read.
// ------------------------------------------------------------------------------------------------
HKD uint64_t left_shift2(uint64_t x) {
    x &= 0xffffffffull;
    x = (x ^ (x << 16)) & 0x0000ffff0000ffffull;
    x = (x ^ (x << 8)) & 0x00ff00ff00ff00ffull;
    x = (x ^ (x << 4)) & 0x0f0f0f0f0f0f0f0full;
    x = (x ^ (x << 2)) & 0x3333333333333333ull;
    x = (x ^ (x << 1)) & 0x5555555555555555ull;
    return x;
}
HKD uint32_t fast_owen_scramble(uint32_t v, uint32_t seed) {
    v = __brev(v);
    v ^= v * 0x3d20adeau;
    v += seed;
    v *= (seed >> 16) | 1u;
    v ^= v * 0x05526c56u;
    v ^= v * 0x53a22864u;
    return __brev(v);
}
// Generator-matrix product for Sobol dimensions 0 and 1 in closed form (hk_ctx_set_tables verifies that the
// caller's SobolMatrices32 really has this structure, otherwise the table loop below is used):
//   dim 0: column b = 0x80000000 >> b (b < 32), 0 beyond          =>  v = bitreverse(low 32 bits of a)
//   dim 1: column b = (1+x)^b over GF(2), period 32 in b          =>  v = bitreverse(zeta(lo)) ^ bitreverse(zeta(hi))
//          where zeta is the superset-sum transform (Lucas: C(b,j) odd iff j is a sub-mask of b).
HKD uint32_t sobol_pascal32(uint32_t a) {
    a ^= (a >> 1) & 0x55555555u;
    a ^= (a >> 2) & 0x33333333u;
    a ^= (a >> 4) & 0x0f0f0f0fu;
    a ^= (a >> 8) & 0x00ff00ffu;
    a ^= (a >> 16) & 0x0000ffffu;
    return __brev(a);
}
HKD uint32_t sobol_matrix_product(uint64_t a, int dimension, const uint32_t* __restrict__ mats) {
    if (mats == nullptr) {  // closed forms (wave-uniform branch)
        uint32_t lo = (uint32_t)a, hi = (uint32_t)(a >> 32);
        if (dimension == 0) return __brev(lo);
        return sobol_pascal32(lo) ^ sobol_pascal32(hi);
    }
    uint32_t v = 0;
    const uint32_t* m = mats + dimension * 52;
#pragma unroll 4
    for (int b = 0; b < 52; ++b) {
        uint32_t mask = 0u - (uint32_t)((a >> b) & 1ull);
        v ^= m[b] & mask;
    }
    return v;
}
HKD float sobol_sample(uint64_t a, int dimension, uint32_t scramble, const uint32_t* __restrict__ mats) {
    uint32_t v = fast_owen_scramble(sobol_matrix_product(a, dimension, mats), scramble);
    float f = (float)v * 2.3283064365386963e-10f;
    const float lim = 1.0f - 1.1920929e-7f;
    return f < lim ? f : lim;
}
// PERMUTATIONS_4WAY packed: 4 x 2 bits per permutation (sobol.jl:155-180):
//    0xE4, 0xB4, 0xD8, 0x78, 0x6C, 0x9C, 0xE1, 0xB1, 0xC9, 0x39, 0x2D, 0x8D, 0xC6, 0x36, 0xD2, 0x72, 0x4E, 0x1E, 0x27, 0x87, 0x1B, 0x4B, 0x63, 0x93
// held as three 64-bit immediates (8 permutations each): as a constant array the lookup is a byte load with a per-lane index, 24
// of them per path vertex, and per-lane cache-line lookups are what these kernels run out of (k_shade -11 % without them).
// Digits i = i_hi .. i_lo of the permuted index (sobol.jl:225-262).  The full index is digits n-1 .. pow2 plus the last-bit
// special case; the digits whose bits lie above the sample bits (shift >= log2_spp) do not depend on the sample index.
// index (0..23) of the digit permutation chosen by the digits above: (mix_bits(higher ^ dmix) >> 24) % 24
HKD int zsobol_perm_index(uint64_t higher, uint64_t dmix) {
    uint64_t h = mix_bits(higher ^ dmix);
    // (h >> 24) % 24 on a 40-bit value with 32-bit arithmetic: 2^32 mod 24 == 16
    uint32_t xlo = (uint32_t)(h >> 24), xhi = (uint32_t)(h >> 56);
    return (int)((xhi * 16u + xlo % 24u) % 24u);
}
HKD uint64_t zsobol_permute_digit(int p, int digit) {
    const uint64_t pw = p < 8 ? 0xb1e19c6c78d8b4e4ull : (p < 16 ? 0x72d236c68d2d39c9ull : 0x93634b1b87271e4eull);
    return (pw >> (8 * (p & 7) + 2 * digit)) & 3ull;
}
HKD uint64_t zsobol_digits(uint64_t morton, uint64_t dmix, int pow2, int i_hi, int i_lo) {
    uint64_t sample_index = 0;
    for (int i = i_hi; i >= i_lo; --i) {
        int shift = 2 * i - pow2;
        int digit = (int)((morton >> shift) & 3ull);
        sample_index |= zsobol_permute_digit(zsobol_perm_index(morton >> (shift + 2), dmix), digit) << shift;
    }
    return sample_index;
}
HKD int zsobol_first_pixel_digit(int log2_spp) { return (log2_spp + 1) >> 1; }  // smallest i with 2i - pow2 >= log2_spp
HKD uint64_t zsobol_sample_index(uint64_t morton, int dimension, int log2_spp, int n_base4_digits) {
    const int pow2 = log2_spp & 1;
    const uint64_t dmix = 0x55555555ull * (uint64_t)(int64_t)dimension;
    // digits i = n-1 .. last_digit; iterations with i < last_digit contribute nothing (sobol.jl:244-247)
    uint64_t sample_index = zsobol_digits(morton, dmix, pow2, n_base4_digits - 1, pow2);
    if (pow2) {
        uint64_t digit = morton & 1ull;
        uint64_t xb = mix_bits((morton >> 1) ^ dmix) & 1ull;
        sample_index |= (digit ^ xb);
    }
    return sample_index;
}
// The permutation of a digit is chosen by a hash of the digits ABOVE it.  For the top sample digit those are the pixel's digits alone;
// for the second one, the pixel's digits and the top sample digit (4 cases): per (pixel, dimension) five permutation indices, 5 bits
// each, are tabulated next to the permuted pixel digits (k_sobol_table), and a draw hashes only the remaining sample digits
// (4 instead of 6 at log2_spp = 12: two 64-bit mix_bits and two mod-24 fewer per draw, five draws per path vertex).
HKD uint32_t zsobol_top_perms(uint64_t morton_pixel_shifted, uint64_t dmix, int log2_spp) {   // morton with zero sample bits
    const int pow2 = log2_spp & 1;
    const int i5 = zsobol_first_pixel_digit(log2_spp) - 1;
    uint32_t packed = 0;
    if (i5 >= pow2) packed = (uint32_t)zsobol_perm_index(morton_pixel_shifted >> (2 * i5 - pow2 + 2), dmix);
    if (i5 - 1 >= pow2) {
        const int shift5 = 2 * i5 - pow2, shift4 = shift5 - 2;
        for (int d5 = 0; d5 < 4; ++d5)
            packed |= (uint32_t)zsobol_perm_index((morton_pixel_shifted | ((uint64_t)d5 << shift5)) >> (shift4 + 2), dmix) << (5 + 5 * d5);
    }
    return packed;
}
// same value as zsobol_sample_index, with the pixel digits (`hi` = their permuted digits >> log2_spp) and the permutation indices
// of the two top sample digits (`perms`, zsobol_top_perms) read from the table entry
HKD uint64_t zsobol_sample_index_cached(uint64_t morton, int dimension, int log2_spp, uint32_t hi, uint32_t perms) {
    const int pow2 = log2_spp & 1;
    const uint64_t dmix = 0x55555555ull * (uint64_t)(int64_t)dimension;
    const int i5 = zsobol_first_pixel_digit(log2_spp) - 1;
    uint64_t sample_index = (uint64_t)hi << log2_spp;
    int next = i5;
    if (i5 >= pow2) {
        const int shift5 = 2 * i5 - pow2;
        const int d5 = (int)((morton >> shift5) & 3ull);
        sample_index |= zsobol_permute_digit((int)(perms & 31u), d5) << shift5;
        next = i5 - 1;
        if (i5 - 1 >= pow2) {
            const int shift4 = shift5 - 2;
            const int d4 = (int)((morton >> shift4) & 3ull);
            sample_index |= zsobol_permute_digit((int)((perms >> (5 + 5 * d5)) & 31u), d4) << shift4;
            next = i5 - 2;
        }
    }
    sample_index |= zsobol_digits(morton, dmix, pow2, next, pow2);
    if (pow2) {
        uint64_t digit = morton & 1ull;
        uint64_t xb = mix_bits((morton >> 1) ^ dmix) & 1ull;
        sample_index |= (digit ^ xb);
    }
    return sample_index;
}
// Dimensions the path draws (volpath.jl:252-262): camera 1,3,4,6 and, per depth d, 6+7d + {1,3,4,6,7}.  Row index into the
// pixel-digit table, or -1 for a dimension that is not tabulated.
HKD int sobol_row(int dim) {
    if (dim < 7) return dim == 1 ? 0 : (dim == 3 ? 1 : (dim == 4 ? 2 : (dim == 6 ? 3 : -1)));
    int d = (dim - 6) / 7, o = (dim - 6) - 7 * d;   // o == 0 is the previous depth's "+7"
    int j = o == 0 ? 0 : (o == 1 ? 1 : (o == 3 ? 2 : (o == 4 ? 3 : (o == 6 ? 4 : -1))));
    return j < 0 ? -1 : 4 + 5 * d + j;
}
HKD int sobol_row_dim(int row) {  // inverse of sobol_row
    if (row < 4) return row == 0 ? 1 : (row == 1 ? 3 : (row == 2 ? 4 : 6));
    int d = (row - 4) / 5, j = (row - 4) - 5 * d;
    return 6 + 7 * d + (j == 0 ? 0 : (j == 1 ? 1 : (j == 2 ? 3 : (j == 3 ? 4 : 6))));
}
HKD uint64_t zsobol_hash(int dimension, uint32_t seed) {
    uint32_t w[2] = {(uint32_t)dimension, seed};
    return murmur64a_words<2>(w);
}
struct SobolCtx {
    // the pixel (1-based) either directly or as its tile-major slot within the rendered range: the Morton code is needed only by
    // the draws that miss the tables, and is formed there (two integer divisions and two bit spreads per path vertex otherwise)
    int px, py;            // 1-based pixel coordinates, or px == 0: derive from pix_slot
    int pix_slot, x0, y0, tiles_x;
    int sample_idx;
    const uint32_t* mats;
    const uint2* hi;       // pixel-digit table column of this pixel (null: compute every digit)
    const uint16_t* lo;    // this path's entry of row 0 in the sample-bit table (null: hash the sample digits)
    int lo_rows;
    uint32_t lo_row_stride;   // entries per row of the sample-bit table (pixel slots * entries per pixel)
    int hi_rows, hi_stride;
    int log2_spp, n_digits;
    uint32_t seed;
};
HKD void sobol_ctx_tables(SobolCtx& c, const DSobol& s, const uint32_t* mats, int sample_idx, int pix_slot, int k) {
    c.sample_idx = sample_idx;
    c.pix_slot = pix_slot;
    c.mats = mats;
    // the table holds the digits above the sample bits: valid only while the sample index fits in them
    const bool cached = s.hi_table != nullptr && pix_slot >= 0 && ((unsigned)sample_idx >> s.log2_spp) == 0u;
    c.hi = cached ? s.hi_table + pix_slot : nullptr;
    const bool low = cached && s.lo_table != nullptr && k >= 0 && (unsigned)(s.lo_offset + k) < (unsigned)s.lo_count;
    c.lo = low ? s.lo_table + ((size_t)pix_slot * s.lo_count + (size_t)(s.lo_offset + k)) : nullptr;
    c.lo_rows = low ? s.lo_rows : 0;
    c.lo_row_stride = (uint32_t)s.hi_stride * (uint32_t)s.lo_count;
    c.hi_rows = s.hi_rows;
    c.hi_stride = s.hi_stride;
    c.log2_spp = s.log2_spp;
    c.n_digits = s.n_base4_digits;
    c.seed = s.seed;
}
// px, py: 1-based pixel coordinates; pix_slot: the pixel's slot within one sample of the pass (tile-major), or -1 when the caller has
// no table column; k: the path's sample within the pass (sample_idx = first_sample + k * stride), or -1
HKD SobolCtx sobol_ctx(const DSobol& s, const uint32_t* mats, int px, int py, int sample_idx, int pix_slot = -1, int k = -1) {
    SobolCtx c;
    c.px = px;
    c.py = py;
    c.x0 = c.y0 = 0;
    c.tiles_x = 1;
    sobol_ctx_tables(c, s, mats, sample_idx, pix_slot, k);
    return c;
}
// the render kernels' form: the pixel as its slot of the rendered range [x0, ..) x [y0, ..) in 8x8 tiles, tiles_x tiles per row
HKD SobolCtx sobol_ctx_slot(const DSobol& s, const uint32_t* mats, int x0, int y0, int tiles_x, int pix_slot, int k, int sample_idx) {
    SobolCtx c;
    c.px = c.py = 0;
    c.x0 = x0;
    c.y0 = y0;
    c.tiles_x = tiles_x;
    sobol_ctx_tables(c, s, mats, sample_idx, pix_slot, k);
    return c;
}
HKD uint64_t sobol_morton_base(const SobolCtx& c) {   // encode_morton2(px, py) << log2_spp | sample_idx
    int px = c.px, py = c.py;
    if (px == 0) {
        const int tile = c.pix_slot >> 6, l = c.pix_slot & 63;
        const int ty = tile / c.tiles_x, tx = tile - ty * c.tiles_x;
        px = c.x0 + tx * 8 + (l & 7) + 1;
        py = c.y0 + ty * 8 + (l >> 3) + 1;
    }
    const uint64_t m = (left_shift2((uint64_t)(uint32_t)py) << 1) | left_shift2((uint64_t)(uint32_t)px);
    return (m << c.log2_spp) | (uint64_t)(int64_t)c.sample_idx;
}
// The draws that miss the sample-bit table (rows beyond the table's budget, one-sample progressive calls, the point-wise test kernels)
// hash the digits in line: moved out of line (noinline) the call made k_shade<0> 3 % and k_track 3 % slower (288 B of scratch for the
// callee's frame) although it removes ~1600 static instructions from every path vertex.
// TABLE_ONLY: the caller knows (from the host: DSobol::lo_rows against the rows a bounce reads) that both tables hold this draw for
// every path of the launch — two loads, and none of the hashing code (and its registers) in the kernel.
template <bool TABLE_ONLY = false>
HKD uint64_t sobol_index(const SobolCtx& c, int dim) {
    const int row = sobol_row(dim);
    if (TABLE_ONLY) return ((uint64_t)c.hi[(size_t)row * c.hi_stride].x << c.log2_spp) | (uint64_t)c.lo[(size_t)row * c.lo_row_stride];
    if (c.hi != nullptr && row >= 0 && row < c.hi_rows) {
        const uint2 e = c.hi[(size_t)row * c.hi_stride];
        if (row < c.lo_rows) return ((uint64_t)e.x << c.log2_spp) | (uint64_t)c.lo[(size_t)row * c.lo_row_stride];
        return zsobol_sample_index_cached(sobol_morton_base(c), dim, c.log2_spp, e.x, e.y);
    }
    return zsobol_sample_index(sobol_morton_base(c), dim, c.log2_spp, c.n_digits);
}
template <bool TABLE_ONLY = false>
HKD float sobol_1d(const SobolCtx& c, int dim) {  // sobol.jl:269-282
    uint64_t idx = sobol_index<TABLE_ONLY>(c, dim);
    uint32_t h = (uint32_t)zsobol_hash(dim + 1, c.seed);
    return sobol_sample(idx, 0, h, c.mats);
}
template <bool TABLE_ONLY = false>
HKD v2 sobol_2d(const SobolCtx& c, int dim) {  // sobol.jl:290-309
    uint64_t idx = sobol_index<TABLE_ONLY>(c, dim);
    uint64_t bits = zsobol_hash(dim + 2, c.seed);
    return mk2(sobol_sample(idx, 0, (uint32_t)bits, c.mats), sobol_sample(idx, 1, (uint32_t)(bits >> 32), c.mats));
}

// ------------------------------------------------------------------------------------------------
// sin / cos of a binary32 argument the way Julia's Base computes them (base/special/trig.jl, a port of FreeBSD msun's k_sinf /
// k_cosf / e_rem_pio2f): argument reduction and the polynomial kernels in binary64, ONE rounding to binary32 at the end.  The
// reference calls Base.sin / Base.cos on Float32, the oracle evaluates the same expressions (oracle/hko_core.h), so the two sides
// agree bit for bit here — and a double-precision kernel is ~25 fp64 operations for both results, against ~330 instructions for
// each of ocml's sinf and cosf.
// ------------------------------------------------------------------------------------------------
HKD float jl_sin_kernel(double y) {
    const double S1 = -0.16666666641626524, S2 = 0.008333329385889463, S3 = -0.00019839334836096632, S4 = 2.718311493989822e-6;
    double z = y * y, w = z * z;
    double r = S3 + z * S4, s = z * y;
    return (float)((y + s * (S1 + z * S2)) + s * w * r);
}
HKD float jl_cos_kernel(double y) {
    const double C0 = -0.499999997251031, C1 = 0.04166662332373906, C2 = -0.001388676377460993, C3 = 2.439044879627741e-5;
    double z = y * y, w = z * z;
    double r = C2 + z * C3;
    return (float)(((1.0 + z * C0) + w * C1) + (w * z) * r);
}
// rem_pio2_kernel(x::Float32): n and the reduced argument (binary64) with x = n * pi/2 + y, |x| < 2^28 * pi/2
HKD int jl_rem_pio2(float x, double& y) {
    const double PI = 3.141592653589793;
    const double xd = (double)x, ax = fabs(xd);
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
    const double fn = rint(xd * 6.36619772367581382433e-01);   // Cody-Waite with a 33 + 53 bit pi/2
    const double r = xd - fn * 1.57079631090164184570e+00, w = fn * 1.58932547735281966916e-08;
    y = r - w;
    return (int)fn;
}
HKD void jl_sincos(float x, float& s, float& c) {
    const float ax = fabsf(x);
    if (ax < 0.7853982f) {   // Float32(pi)/4: no reduction
        s = ax < 0.00034526698f ? x : jl_sin_kernel((double)x);      // sqrt(eps(Float32))
        c = ax < 0.00024414062f ? 1.0f : jl_cos_kernel((double)x);   // sqrt(eps(Float32)/2)
        return;
    }
    if (!(ax < 2.1e8f)) {   // beyond the medium range (never reached by the sampling code): libm
        s = sinf(x);
        c = cosf(x);
        return;
    }
    double y;
    const int n = jl_rem_pio2(x, y) & 3;
    const float sk = jl_sin_kernel(y), ck = jl_cos_kernel(y);
    s = n == 0 ? sk : (n == 1 ? ck : (n == 2 ? -sk : -ck));
    c = n == 0 ? ck : (n == 1 ? -sk : (n == 2 ? -ck : sk));
}

// ------------------------------------------------------------------------------------------------
// sampling primitives (sampler/sampling.jl:1-30)
// ------------------------------------------------------------------------------------------------
HKD v2 concentric_sample_disk(v2 u) {
    float ox = 2.0f * u.x - 1.0f, oy = 2.0f * u.y - 1.0f;
    float sx = ox + 1.0e-10f, sy = oy + 1.0e-10f;
    bool xl = fabsf(ox) > fabsf(oy);
    float r = xl ? ox : oy;
    float theta = xl ? ((oy / sx) * PI_F) / 4.0f : PI_F / 2.0f - ((ox / sy) * PI_F) / 4.0f;
    float st, ct;
    jl_sincos(theta, st, ct);
    return mk2(r * ct, r * st);
}
HKD v3 cosine_sample_hemisphere(v2 u) {
    v2 d = concentric_sample_disk(u);
    float z = sqrtf(maxf(0.0f, 1.0f - d.x * d.x - d.y * d.y));
    return mk3(d.x, d.y, z);
}

// ------------------------------------------------------------------------------------------------
// spectral (spectral/spectral.jl:192-249, rgb2spec.jl, uplift.jl, color.jl)
// ------------------------------------------------------------------------------------------------
// atanh and cosh over the ranges the wavelength sampler uses (|x| < 0.971; |y| < 2.2), through the hardware's exp2 / log2 / rcp
// (1 ulp each): 0.5 ln((1+x)/(1-x)) and (e^y + e^-y)/2.  Absolute error ~1e-7 in atanh = 1.4e-5 nm in the wavelength (a quarter
// of a float's spacing at 540 nm), ~3e-7 relative in the pdf: as close to the reference's libm as ocml's atanhf / coshf were
// (the oracle's glibc is a third implementation), at 12 instructions instead of ~110 — a third of k_camera's arithmetic.
HKD float atanh_sampler(float x) { return 0.34657359027997264f * __builtin_amdgcn_logf((1.0f + x) * __builtin_amdgcn_rcpf(1.0f - x)); }
HKD float cosh_sampler(float y) {
    const float e = __builtin_amdgcn_exp2f(1.4426950408889634f * y);
    return 0.5f * (e + __builtin_amdgcn_rcpf(e));
}
HKD float visible_wavelengths_pdf(float l) {
    if (l < 360.0f || l > 830.0f) return 0.0f;
    float c = cosh_sampler(0.0072f * (l - 538.0f));
    return 0.0039398042f / (c * c);
}
HKD float sample_visible_wavelength(float u) { return 538.0f - 138.888889f * atanh_sampler(0.85691062f - 1.82750197f * u); }
HKD void sample_wavelengths_visible(float u, S4& lambda, S4& pdf) {
    float u2 = u + 0.25f;
    u2 = u2 >= 1.0f ? u2 - 1.0f : u2;
    float u3 = u + 0.5f;
    u3 = u3 >= 1.0f ? u3 - 1.0f : u3;
    float u4 = u + 0.75f;
    u4 = u4 >= 1.0f ? u4 - 1.0f : u4;
    lambda = s4(sample_visible_wavelength(u), sample_visible_wavelength(u2), sample_visible_wavelength(u3), sample_visible_wavelength(u4));
    pdf = s4(visible_wavelengths_pdf(lambda.x), visible_wavelengths_pdf(lambda.y), visible_wavelengths_pdf(lambda.z), visible_wavelengths_pdf(lambda.w));
}
HKD float sigmoidf(float x) {
    if (isinf(x)) return x > 0 ? 1.0f : 0.0f;
    return 0.5f + x / (2.0f * sqrtf(1.0f + x * x));
}
HKD float poly_eval(float c0, float c1, float c2, float l) { return sigmoidf(c0 * l * l + c1 * l + c2); }
HKD S4 poly_eval4(float4 c, S4 l) { return s4(poly_eval(c.x, c.y, c.z, l.x), poly_eval(c.x, c.y, c.z, l.y), poly_eval(c.x, c.y, c.z, l.z), poly_eval(c.x, c.y, c.z, l.w)); }
HKD float poly_max_value(float c0, float c1, float c2) {
    float r = maxf(poly_eval(c0, c1, c2, 360.0f), poly_eval(c0, c1, c2, 830.0f));
    if (c0 != 0) {
        float lc = -c1 / (2.0f * c0);
        if (360.0f <= lc && lc <= 830.0f) r = maxf(r, poly_eval(c0, c1, c2, lc));
    }
    return r;
}
// run-time table lookup (textured colours only; constant colours are baked on the host)
template <bool TWO_PLANES = false>
HKD void rgb_to_spectrum(const DTables& T, float r, float g, float b, float& c0, float& c1, float& c2) {
    r = clampf(r, 0.0f, 1.0f);
    g = clampf(g, 0.0f, 1.0f);
    b = clampf(b, 0.0f, 1.0f);
    if (r == g && g == b) {
        c0 = c1 = 0.0f;
        c2 = (r > 0.0f && r < 1.0f) ? (r - 0.5f) / sqrtf(r * (1.0f - r)) : (r <= 0.0f ? -1.0e10f : 1.0e10f);
        return;
    }
    int maxc = r > g ? (r > b ? 1 : 3) : (g > b ? 2 : 3);
    float z = maxc == 1 ? r : (maxc == 2 ? g : b);
    float xc = maxc == 1 ? g : (maxc == 2 ? b : r);
    float yc = maxc == 1 ? b : (maxc == 2 ? r : g);
    int res = T.rgb2spec_res;
    float x = xc * (float)(res - 1) / z, y = yc * (float)(res - 1) / z;
    // zi = the largest i in 1 .. res-1 with scale[i-1] < z (1 when there is none): the reference's linear scan (rgb2spec.jl:118-124).
    // The scale table is non-decreasing (hk_ctx_set_tables checks it), so the entries below z form a prefix: 6 probes, not 63.
    int zi;
    if (T.rgb2spec_sorted) {
        int lo = 0, hi = res - 1;   // first j in [0, res-1) with !(scale[j] < z)
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (T.rgb2spec_scale[mid] < z)
                lo = mid + 1;
            else
                hi = mid;
        }
        zi = lo < 1 ? 1 : lo;
    } else {
        zi = 1;
        for (int i = 1; i <= res - 1; ++i)
            if (T.rgb2spec_scale[i - 1] < z) zi = i;
    }
    zi = zi < res - 1 ? zi : res - 1;
    int xi = (int)x + 1, yi = (int)y + 1;
    xi = xi < res - 1 ? xi : res - 1;
    yi = yi < res - 1 ? yi : res - 1;
    float dx = x - (float)(xi - 1), dy = y - (float)(yi - 1);
    float dz = (z - T.rgb2spec_scale[zi - 1]) / (T.rgb2spec_scale[zi] - T.rgb2spec_scale[zi - 1]);
    // the eight corners as float4 grid points (c0, c1, c2, -), x fastest (DTables::rgb2spec_points): 8 per-lane loads instead of 24
    const size_t R = (size_t)res;
    const float4* __restrict__ P = T.rgb2spec_points + (((size_t)(maxc - 1) * R + (size_t)(zi - 1)) * R + (size_t)(yi - 1)) * R + (size_t)(xi - 1);
    float out[3];
    if (TWO_PLANES) {
        // One z plane at a time (4 corners = 16 registers live, not 32): the second plane's address is made to depend on the first
        // plane's result, otherwise the compiler hoists all eight loads.  Used by the Matte reflectance only: with all eight in
        // flight k_shade<Matte> spills into its hot path (Cornell, which never comes here, +4 %); the other kinds run faster with
        // the eight loads in flight (sky: k_shade -15 % against the two-plane form).
        float pl[2][3];
        unsigned plane = 0;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const float4 a00 = P[plane], a01 = P[plane + 1], a10 = P[plane + R], a11 = P[plane + R + 1];
#define HK_R2S_LERP(F) ((1.0f - dy) * ((1.0f - dx) * a00.F + dx * a01.F) + dy * ((1.0f - dx) * a10.F + dx * a11.F))
            pl[h][0] = HK_R2S_LERP(x);
            pl[h][1] = HK_R2S_LERP(y);
            pl[h][2] = HK_R2S_LERP(z);
#undef HK_R2S_LERP
            plane = (unsigned)(res * res);
            if (h == 0) asm volatile("" : "+v"(plane) : "v"(pl[0][0]), "v"(pl[0][1]), "v"(pl[0][2]));
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) out[k] = (1.0f - dz) * pl[0][k] + dz * pl[1][k];
    } else {
        const float4 a000 = P[0], a001 = P[1], a010 = P[R], a011 = P[R + 1], a100 = P[R * R], a101 = P[R * R + 1], a110 = P[R * R + R], a111 = P[R * R + R + 1];
#define HK_R2S_LERP(F)                                                                                                                  \
    ((1.0f - dz) * ((1.0f - dy) * ((1.0f - dx) * a000.F + dx * a001.F) + dy * ((1.0f - dx) * a010.F + dx * a011.F)) +                    \
     dz * ((1.0f - dy) * ((1.0f - dx) * a100.F + dx * a101.F) + dy * ((1.0f - dx) * a110.F + dx * a111.F)))
        out[0] = HK_R2S_LERP(x);
        out[1] = HK_R2S_LERP(y);
        out[2] = HK_R2S_LERP(z);
#undef HK_R2S_LERP
    }
    c0 = out[0];
    c1 = out[1];
    c2 = out[2];
}
__device__ static const float kD65[107] = {
    0.0341f,  1.6643f,  3.2945f,  11.7652f, 20.236f,  28.6447f, 37.0535f, 38.5011f, 39.9488f, 42.4302f, 44.9117f, 45.775f,
    46.6383f, 49.3637f, 52.0891f, 51.0323f, 49.9755f, 52.3118f, 54.6482f, 68.7015f, 82.7549f, 87.1204f, 91.486f,  92.4589f,
    93.4318f, 90.057f,  86.6823f, 95.7736f, 104.865f, 110.936f, 117.008f, 117.41f,  117.812f, 116.336f, 114.861f, 115.392f,
    115.923f, 112.367f, 108.811f, 109.082f, 109.354f, 108.578f, 107.802f, 106.296f, 104.79f,  106.239f, 107.689f, 106.047f,
    104.405f, 104.225f, 104.046f, 102.023f, 100.0f,   98.1671f, 96.3342f, 96.0611f, 95.788f,  92.2368f, 88.6856f, 89.3459f,
    90.0062f, 89.8026f, 89.5991f, 88.6489f, 87.6987f, 85.4936f, 83.2886f, 83.4939f, 83.6992f, 81.863f,  80.0268f, 80.1207f,
    80.2146f, 81.2462f, 82.2778f, 80.281f,  78.2842f, 74.0027f, 69.7213f, 70.6652f, 71.6091f, 72.979f,  74.349f,  67.9765f,
    61.604f,  65.7448f, 69.8856f, 72.4863f, 75.087f,  69.3398f, 63.5927f, 55.0054f, 46.4182f, 56.6118f, 66.8054f, 65.0941f,
    63.3828f, 63.8434f, 64.304f,  61.8779f, 59.4519f, 55.7054f, 51.959f,  54.6998f, 57.4406f, 58.8765f, 60.3125f};
HKD float sample_d65(float l) {  // uplift.jl:437-457
    if (l <= 300.0f) return kD65[0];
    if (l >= 830.0f) return kD65[106];
    float t = (l - 300.0f) / 5.0f;
    float fl = floorf(t);
    int idx = clampi((int)fl + 1, 1, 106);
    float frac = t - fl;
    return kD65[idx - 1] * (1.0f - frac) + kD65[idx] * frac;
}
HKD S4 d65_4(S4 l) { return s4(sample_d65(l.x), sample_d65(l.y), sample_d65(l.z), sample_d65(l.w)); }

// the three uplifts, split into "coefficients" (host-baked or table lookup) and "evaluate"
template <bool TWO_PLANES = false>
HKD float4 coef_bounded(const DTables& T, float r, float g, float b) {  // uplift_rgb
    float c0, c1, c2;
    rgb_to_spectrum<TWO_PLANES>(T, r, g, b, c0, c1, c2);
    return make_float4(c0, c1, c2, 1.0f);
}
template <bool TWO_PLANES = false>
HKD float4 coef_unbounded(const DTables& T, float r, float g, float b) {  // uplift_rgb_unbounded (Q5)
    float m = maxf(maxf(r, g), b);
    if (m <= 0.0f) return make_float4(0.0f, 0.0f, 0.0f, 0.0f);  // scale 0 => zero spectrum
    float c0, c1, c2;
    rgb_to_spectrum<TWO_PLANES>(T, r / m, g / m, b / m, c0, c1, c2);
    return make_float4(c0, c1, c2, m / poly_max_value(c0, c1, c2));
}
template <bool TWO_PLANES = false>
HKD float4 coef_illuminant(const DTables& T, float r, float g, float b) {  // rgb_to_spectral_sigmoid_illuminant
    float m = maxf(maxf(r, g), b);
    if (m <= 0.0f) return make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    float scale = 2.0f * m;
    float c0, c1, c2;
    rgb_to_spectrum<TWO_PLANES>(T, r / scale, g / scale, b / scale, c0, c1, c2);
    return make_float4(c0, c1, c2, scale);
}
HKD S4 eval_bounded(float4 c, S4 l) { return poly_eval4(c, l); }
HKD S4 eval_scaled(float4 c, S4 l) {  // scale * poly(lambda); scale == 0 encodes the m <= 0 early-out
    if (c.w == 0.0f) return s4(0.0f);
    S4 p = poly_eval4(c, l);
    return s4(c.w * p.x, c.w * p.y, c.w * p.z, c.w * p.w);
}
HKD S4 eval_illuminant(float4 c, S4 l) {  // scale * poly(lambda) * D65(lambda)
    if (c.w == 0.0f) return s4(0.0f);
    S4 p = poly_eval4(c, l);
    S4 d = d65_4(l);
    return s4(c.w * p.x * d.x, c.w * p.y * d.y, c.w * p.z * d.z, c.w * p.w * d.w);
}

HKD float sample_cie(const float* tab, float l) {
    int off = (int)rintf(l) - 360;
    if (off < 0 || off >= 471) return 0.0f;
    return tab[off];
}
HKD v3 spectral_to_rgb_clamped(const DTables& T, S4 L, S4 lambda, S4 pdf, float max_component_value) {  // volpath.jl:343-361
    v3 sum = mk3(0.0f, 0.0f, 0.0f);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float p = at(pdf, i);
        if (p != 0.0f) {
            float l = at(lambda, i), Li = at(L, i);
            v3 cmf = mk3(sample_cie(T.cie, l), sample_cie(T.cie + 471, l), sample_cie(T.cie + 942, l));
            sum = sum + (cmf * Li) / p;
        }
    }
    v3 xyz = sum * 0.25f;
    float X = xyz.x, Y = xyz.y, Z = xyz.z;
    v3 rgb = mk3(3.2404542f * X - 1.5371385f * Y - 0.4985314f * Z, -0.9692660f * X + 1.8760108f * Y + 0.0415560f * Z,
                 0.0556434f * X - 0.2040259f * Y + 1.0572252f * Z);
    rgb = mk3(maxf(0.0f, rgb.x), maxf(0.0f, rgb.y), maxf(0.0f, rgb.z));
    float m = maxf(maxf(rgb.x, rgb.y), rgb.z);
    if (m > max_component_value) rgb = rgb * (max_component_value / m);
    return rgb;
}

// ------------------------------------------------------------------------------------------------
// textures (textures/texture-ref.jl:151-186)
// ------------------------------------------------------------------------------------------------
HKD void tex_bilinear(const DTexture& t, v2 uv, float out[4]) {
    float ua0 = 1.0f - uv.y, ua1 = uv.x;
    int h = t.height, w = t.width, ch = t.channels;
    float px = ua1 * (float)(w - 1) + 1.0f, py = ua0 * (float)(h - 1) + 1.0f;
    float fpx = floorf(px), fpy = floorf(py);
    int x0 = (int)fpx, y0 = (int)fpy;
    int x1 = clampi(x0 + 1, 1, w), y1 = clampi(y0 + 1, 1, h);
    x0 = clampi(x0, 1, w);
    y0 = clampi(y0, 1, h);
    float fx = px - fpx, fy = py - fpy;
    const float* p00 = t.data + ((size_t)(y0 - 1) + (size_t)h * (size_t)(x0 - 1)) * ch;
    const float* p10 = t.data + ((size_t)(y0 - 1) + (size_t)h * (size_t)(x1 - 1)) * ch;
    const float* p01 = t.data + ((size_t)(y1 - 1) + (size_t)h * (size_t)(x0 - 1)) * ch;
    const float* p11 = t.data + ((size_t)(y1 - 1) + (size_t)h * (size_t)(x1 - 1)) * ch;
    // channels is 1 or 4: a loop of four with the channel test inside unrolls to constant indices, so `out` stays in registers (a loop
    // to a run-time bound made the compiler keep every caller's float[4] in LDS)
#pragma unroll
    for (int c = 0; c < 4; ++c)
        if (c < ch) {
            float c0 = p00[c] * (1.0f - fx) + p10[c] * fx;
            float c1 = p01[c] * (1.0f - fx) + p11[c] * fx;
            out[c] = c0 * (1.0f - fy) + c1 * fy;
        }
}
// eval_tex(ctx, ref, uv::Point2f) == _sample_texture_data (textures/basic.jl:19-26): NEAREST texel by truncation
HKD float eval_f32_nearest(const DScene& sc, const DMaterial& m, int slot, v2 uv) {
    if (m.ftex[slot] < 0) return m.f[slot];
    const DTexture& t = sc.textures[m.ftex[slot]];
    if (t.pad == 1) return 0.5f;
    int i = clampi((int)(1.0f + (float)(t.height - 1) * (1.0f - uv.y)), 1, t.height);
    int j = clampi((int)(1.0f + (float)(t.width - 1) * uv.x), 1, t.width);
    return t.data[((size_t)(i - 1) + (size_t)t.height * (size_t)(j - 1)) * t.channels];
}
// TextureFilterContext as far as the path reads it (texture-ref.jl:21-29): uv + (face_idx, bary) for vertex-colour textures.
// A plain uv converts implicitly (face 0 = "no face": vertex-colour textures then read their gray placeholder, :245).
struct TexCtx {
    v2 uv;
    uint32_t face;   // TriangleMeta.primitive_index (1-based face in its mesh)
    float bu, bv;    // barycentrics of vertices 1 and 2; vertex 0 gets w = 1 - bu - bv
    HKD TexCtx(v2 u) : uv(u), face(0u), bu(0.0f), bv(0.0f) {}
    HKD TexCtx(v2 u, uint32_t f, float a, float b) : uv(u), face(f), bu(a), bv(b) {}
};
// eval_tex(ctx, tex, tfc): image textures are bilinear (texture-ref.jl:71-74, 151-186); a VertexColorTexture interpolates the
// three colours of face `face_idx` with (w, u, v) (texture-ref.jl:230-235).  kind rides in DTexture.pad.
HKD void tex_bilinear(const DTexture& t, const TexCtx& tc, float out[4]) {
    if (t.pad == 1) {
        if (tc.face == 0u) {  // UV-only evaluation: gray placeholder RGBSpectrum(0.5f0)
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (c < t.channels) out[c] = c < 3 ? 0.5f : 1.0f;
            return;
        }
        const float* f = t.data + (size_t)(tc.face - 1u) * 3u * (size_t)t.channels;
        float w = 1.0f - tc.bu - tc.bv;
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if (c < t.channels) out[c] = f[c] * w + f[t.channels + c] * tc.bu + f[2 * t.channels + c] * tc.bv;
        return;
    }
    tex_bilinear(t, tc.uv, out);
}
// NO_TEX: the scene holds no texture at all (DScene::simple_lights): the lookup is compiled out
template <bool NO_TEX = false>
HKD float eval_f32(const DScene& sc, const DMaterial& m, int slot, const TexCtx& uv) {
    if (NO_TEX || m.ftex[slot] < 0) return m.f[slot];
    float o[4] = {0, 0, 0, 0};
    tex_bilinear(sc.textures[m.ftex[slot]], uv, o);
    return o[0];
}

// ------------------------------------------------------------------------------------------------
// pixel filter (filter.jl:733-953) and camera (camera/perspective.jl:95-128)
// ------------------------------------------------------------------------------------------------
// The reference's search is 20 branchless halvings whatever the table size (filter.jl:876-953).  Once hi - lo == 1 a further step
// re-tests cdf[lo - 1] <= u — true by construction (cdf[0] = 0 <= u, or lo was set by a passed test) — and changes nothing, so the
// loop may stop after ceil(log2(n)) steps with the identical index: 6 dependent table loads instead of 20 for the 48-entry filter
// tables.  n is wave-uniform (kernel argument), so is the trip count.
HKD int find_interval20(const float* cdf, float u, int n) {
    int lo = 1, hi = n + 1;
    const int steps = n <= 1 ? 0 : (32 - __builtin_clz((unsigned)(n - 1)));
    for (int k = 0; k < steps && k < 20; ++k) {
        int mid = (lo + hi) >> 1;
        bool c = cdf[mid - 1] <= u;
        lo = c ? mid : lo;
        hi = c ? hi : mid;
    }
    return lo;
}
HKD float sample_tent(float u, float r) {
    if (u < 0.5f) return -r + r * sqrtf(2.0f * u);
    return r * (1.0f - sqrtf(2.0f * (1.0f - u)));
}
HKD void filter_sample(const DFilter& f, v2 u, float& px, float& py, float& weight) {
    if (f.type == HK_FILTER_BOX) {
        px = lerpf(-f.rx, f.rx, u.x);
        py = lerpf(-f.ry, f.ry, u.y);
        weight = 1.0f;
        return;
    }
    if (f.type == HK_FILTER_TRIANGLE) {
        px = sample_tent(u.x, f.rx);
        py = sample_tent(u.y, f.ry);
        weight = 1.0f;
        return;
    }
    int ny = f.ny, nx = f.nx;
    int o = clampi(find_interval20(f.marginal_cdf, u.y, ny), 1, ny);
    float du = u.y - f.marginal_cdf[o - 1];
    float diff = f.marginal_cdf[o] - f.marginal_cdf[o - 1];
    du = diff > 0.0f ? du / diff : 0.0f;
    float pdf_y = f.func_integral > 0.0f ? f.marginal_func[o - 1] / f.func_integral : 0.0f;
    py = lerpf(f.dmin_y, f.dmax_y, ((float)(o - 1) + du) / (float)ny);
    float row_integral = f.marginal_func[o - 1];
    const float* row = f.conditional_cdf + (size_t)(o - 1) * (nx + 1);
    int ox = clampi(find_interval20(row, u.x, nx), 1, nx);
    float dux = u.x - row[ox - 1];
    float diffx = row[ox] - row[ox - 1];
    dux = diffx > 0.0f ? dux / diffx : 0.0f;
    float fval = f.func[(size_t)(o - 1) * nx + (ox - 1)];
    float pdf_x = row_integral > 0.0f ? fval / row_integral : 0.0f;
    px = lerpf(f.dmin_x, f.dmax_x, ((float)(ox - 1) + dux) / (float)nx);
    float pdf = pdf_x * pdf_y;
    weight = pdf > 0.0f ? fval / pdf : 0.0f;
}
HKD v3 xform_point(const float* m, v3 p) {
    float x = m[0] * p.x + m[1] * p.y + m[2] * p.z + m[3];
    float y = m[4] * p.x + m[5] * p.y + m[6] * p.z + m[7];
    float z = m[8] * p.x + m[9] * p.y + m[10] * p.z + m[11];
    float w = m[12] * p.x + m[13] * p.y + m[14] * p.z + m[15];
    if (w == 1.0f) return mk3(x, y, z);
    float inv = 1.0f / w;
    return mk3(x * inv, y * inv, z * inv);
}
HKD v3 xform_vector(const float* m, v3 v) {
    return mk3(m[0] * v.x + m[1] * v.y + m[2] * v.z, m[4] * v.x + m[5] * v.y + m[6] * v.z, m[8] * v.x + m[9] * v.y + m[10] * v.z);
}
HKD void generate_ray(const DCamera& cam, v2 film, v2 lens, float time_u, v3& ro, v3& rd, float& time) {
    v3 pc = xform_point(cam.r2c, mk3(film.x, film.y, 0.0f));
    v3 o = mk3(0.0f, 0.0f, 0.0f);
    v3 d = normalize(pc);
    if (cam.lens_radius > 0) {
        v2 dsk = concentric_sample_disk(lens);
        float plx = cam.lens_radius * dsk.x, ply = cam.lens_radius * dsk.y;
        float t = -cam.focal_distance / d.z;
        v3 pf = o + d * t;
        o = mk3(plx, ply, 0.0f);
        d = normalize(pf - o);
    }
    time = lerpf(cam.shutter_open, cam.shutter_close, time_u);
    ro = xform_point(cam.c2w, o);
    rd = normalize(xform_vector(cam.c2w, d));
}

// ------------------------------------------------------------------------------------------------
// triangle intersection: the arithmetic DESIGN.md "Intersection arithmetic" fixes (== oracle/hko_accel.h)
// ------------------------------------------------------------------------------------------------
HKD bool intersect_triangle(v3 o, v3 d, float t_max, v3 v0, v3 e1, v3 e2, float& t, float& u, float& v) {
    v3 p = cross(d, e2);
    float det = dot(e1, p);
    if (det == 0.0f) return false;
    float inv = 1.0f / det;
    v3 s = o - v0;
    u = dot(s, p) * inv;
    if (!(u >= 0.0f) || u > 1.0f) return false;
    v3 q = cross(s, e1);
    v = dot(d, q) * inv;
    if (!(v >= 0.0f) || u + v > 1.0f) return false;
    t = dot(e2, q) * inv;
    return t > 0.0f && t < t_max;
}

struct HitRec {
    float t;
    int prim;  // original triangle index, -1 = miss
    float u, v;
};

// Per-ray constants of the slab test in FMA form t = b*inv_d + (-o*inv_d).  inv_d is clamped to +-1e30 so axis-parallel rays give
// +-huge instead of inf - inf = NaN (the test only culls; 1e30 is beyond any scene scale).  Rounding slack of the FMA form:
// |err(t)| <= 2^-24 (|o*inv_d| + |t|).  Culling must never drop a triangle whose computed t ties the current best (coplanar
// faces of abutting boxes), so the interval is widened by an absolute term from |o*inv_d| plus a relative 1e-5 on either end;
// the whole slack sits on the far side: n <= f * (1 + 3e-5) + 3 eps admits everything n * 0.99999 - eps <= f * 1.00001 + eps
// admits (n >= 0), in one fma per child.
struct RaySlab {
    float ix, iy, iz, ox, oy, oz, eps3;
};
HKD RaySlab ray_slab(v3 o, v3 d) {
    RaySlab r;
    r.ix = clampf(1.0f / d.x, -1e30f, 1e30f), r.iy = clampf(1.0f / d.y, -1e30f, 1e30f), r.iz = clampf(1.0f / d.z, -1e30f, 1e30f);
    r.ox = -o.x * r.ix, r.oy = -o.y * r.iy, r.oz = -o.z * r.iz;
    r.eps3 = 3.0f * (2.4e-7f * fmaxf(fmaxf(fabsf(r.ix) < 1e30f ? fabsf(r.ox) : 0.0f, fabsf(r.iy) < 1e30f ? fabsf(r.oy) : 0.0f), fabsf(r.iz) < 1e30f ? fabsf(r.oz) : 0.0f));
    return r;
}
// the same ray against QUANTISED nodes (DQNode): planes are grid coordinates q with plane = base + q * cell, so
// t = plane * i + o' = q * (cell * i) + (base * i + o') — the slab of a ray in grid space, one fma per plane as before.  (The rounding of
// the two folded constants moves a plane by ~1e-7 of the scene's size; the boxes carry a whole grid cell, 1.5e-5 of it, of margin.)
HKD RaySlab ray_slab_grid(const DScene& sc, v3 o, v3 d) {
    RaySlab r = ray_slab(o, d);
    r.ox = fmaf(sc.q_base[0], r.ix, r.ox), r.oy = fmaf(sc.q_base[1], r.iy, r.oy), r.oz = fmaf(sc.q_base[2], r.iz, r.oz);
    r.ix *= sc.q_cell[0], r.iy *= sc.q_cell[1], r.iz *= sc.q_cell[2];
    return r;
}
typedef float hk_f2 __attribute__((ext_vector_type(2)));
// One inner-node step of a lane: both child boxes (DNode: the (lo, hi) pair of an axis sits in adjacent words, so one packed fma
// gives both plane distances), nearest hit child first, the other pushed.  Per-lane stack in LDS: entry e of lane l at
// stack[e * 64 + l] (bank = lane => conflict-free).  Selects instead of a four-way branch: a divergent wave would walk every arm.
// The first `nc` nodes (breadth-first order: the top of the tree) may sit in LDS, one array per 16-byte part of the node so that the
// lanes of a wave, each at its own node, spread over all banks: part j of node i at box[j * NC + i], the child pair at child[i].
// (LDS-qualified pointers, and an asm barrier in the LDS arm of node_step: left alone the compiler turns the two arms into a SELECT of
// generic pointers and one flat load.)
typedef float hk_f4v __attribute__((ext_vector_type(4)));
typedef int hk_i2v __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) hk_f4v lds_float4;
typedef __attribute__((address_space(3))) hk_i2v lds_int2;
struct NodeCache {
    const lds_float4* box;
    const lds_int2* child;
    int nc;
    const lds_float4* tri;   // leaf triangles 0 .. nt-1 (media kernels: scenes of a few dozen triangles live in LDS entirely), part j of
    int nt;                  // triangle i at tri[j * NT + i]
};
template <int NC = 0, bool QN = false>
HKD void node_step(const DScene& sc, const RaySlab& rs, float t_best, int* __restrict__ stack, int lane, int& cur, int& sp, const NodeCache& cache = NodeCache()) {
    float4 A, B, C;
    int c0, c1;
    if (QN && !(NC > 0 && cur < cache.nc)) {   // quantised node: two 16-B loads, planes in grid coordinates (rs is the ray's slab in grid space; the LDS copies hold grid coordinates too)
        const uint4* qp = reinterpret_cast<const uint4*>(sc.qnodes) + 2 * (size_t)cur;
        const uint4 P = qp[0], Q = qp[1];
        A = make_float4((float)(P.x & 0xffffu), (float)(P.x >> 16), (float)(P.y & 0xffffu), (float)(P.y >> 16));
        B = make_float4((float)(P.z & 0xffffu), (float)(P.z >> 16), (float)(P.w & 0xffffu), (float)(P.w >> 16));
        C = make_float4((float)(Q.x & 0xffffu), (float)(Q.x >> 16), (float)(Q.y & 0xffffu), (float)(Q.y >> 16));
        c0 = (int)Q.z, c1 = (int)Q.w;
    } else if (NC > 0 && cur < cache.nc) {
        const hk_f4v a = cache.box[cur], b = cache.box[NC + cur], c = cache.box[2 * NC + cur];
        const hk_i2v ch = cache.child[cur];
        A = make_float4(a.x, a.y, a.z, a.w), B = make_float4(b.x, b.y, b.z, b.w), C = make_float4(c.x, c.y, c.z, c.w);
        c0 = ch.x, c1 = ch.y;
        asm volatile("" : "+v"(c0));
    } else {
        const float4* np = reinterpret_cast<const float4*>(sc.nodes) + 4 * (size_t)cur;
        A = np[0], B = np[1], C = np[2];
        const float4 D = np[3];
        c0 = __float_as_int(D.x), c1 = __float_as_int(D.y);
    }
    const hk_f2 IX = {rs.ix, rs.ix}, IY = {rs.iy, rs.iy}, IZ = {rs.iz, rs.iz}, OX = {rs.ox, rs.ox}, OY = {rs.oy, rs.oy}, OZ = {rs.oz, rs.oz};
    const hk_f2 x0 = __builtin_elementwise_fma((hk_f2){A.x, A.y}, IX, OX), y0 = __builtin_elementwise_fma((hk_f2){A.z, A.w}, IY, OY);
    const hk_f2 z0 = __builtin_elementwise_fma((hk_f2){B.x, B.y}, IZ, OZ), x1 = __builtin_elementwise_fma((hk_f2){B.z, B.w}, IX, OX);
    const hk_f2 y1 = __builtin_elementwise_fma((hk_f2){C.x, C.y}, IY, OY), z1 = __builtin_elementwise_fma((hk_f2){C.z, C.w}, IZ, OZ);
    const float n0 = fmaxf(fmaxf(fminf(x0.x, x0.y), fminf(y0.x, y0.y)), fmaxf(fminf(z0.x, z0.y), 0.0f));
    const float f0 = fminf(fminf(fmaxf(x0.x, x0.y), fmaxf(y0.x, y0.y)), fminf(fmaxf(z0.x, z0.y), t_best));
    const float n1 = fmaxf(fmaxf(fminf(x1.x, x1.y), fminf(y1.x, y1.y)), fmaxf(fminf(z1.x, z1.y), 0.0f));
    const float f1 = fminf(fminf(fmaxf(x1.x, x1.y), fmaxf(y1.x, y1.y)), fminf(fmaxf(z1.x, z1.y), t_best));
    const bool h0 = n0 <= fmaf(f0, 1.00003f, rs.eps3);
    const bool h1 = n1 <= fmaf(f1, 1.00003f, rs.eps3);
    const bool both = h0 && h1, any = h0 || h1;
    const bool first0 = n0 <= n1;
    const bool pick0 = h0 & (!h1 | first0);   // and / or of lane masks (scalar): a select between two conditions would be done per lane, && would branch
    const int near_c = pick0 ? c0 : c1;
    if (both) stack[sp * 64 + lane] = pick0 ? c1 : c0;
    sp += both ? 1 : 0;
    const bool pop = !any && sp > 0;
    sp -= pop ? 1 : 0;
    int popped = (int)0x80000000;
    if (pop) popped = stack[sp * 64 + lane];
    cur = any ? near_c : popped;
}

// MODE 0: closest hit (ties on t -> smaller prim index).  MODE 1: shadow segment — returns as soon as an
// opaque triangle is hit (any opaque hit zeroes the contribution, intersection.jl:378-379), otherwise
// closest hit among the non-opaque ones.
template <int MODE, bool COUNT, int NC = 0, int NT = 0>
HKD HitRec traverse(const DScene& sc, v3 o, v3 d, float t_max, int* __restrict__ stack, int lane, unsigned& n_nodes, unsigned& n_tris, bool& opaque_hit,
                    const NodeCache& cache = NodeCache()) {
    HitRec best;
    best.t = t_max;
    best.prim = -1;
    best.u = best.v = 0.0f;
    opaque_hit = false;
    if (sc.n_tris == 0) return best;
    const RaySlab rs = ray_slab(o, d);
    int sp = 0;
    int cur = sc.root_ref;
    const int DONE = (int)0x80000000;
    // "while-while" traversal: every lane first walks inner nodes until it holds a leaf (lanes that already hold
    // one idle), then the wave tests leaves together.  Mixing both per iteration makes a divergent wave pay the
    // node code AND the 4-triangle leaf loop on every step.
    while (cur != DONE) {
        for (;;) {
            const unsigned long long in_nodes = __ballot(cur >= 0);
            if (in_nodes == 0ull) break;
            // stragglers: fewer lanes still descending than waiting with a leaf -> leaves first (see lane_ray_round)
            if (__popcll(in_nodes) < __popcll(__ballot(cur < 0))) break;
            if (cur >= 0) {
                if (COUNT) ++n_nodes;
                node_step<NC>(sc, rs, best.t, stack, lane, cur, sp, cache);
            }
        }
        if (cur < 0 && cur != DONE) {
            int ref = ~cur;
            int first = ref >> 3, count = (ref & 7) + 1;
            for (int i = 0; i < count; ++i) {
                float4 T0, T1, T2;
                if (NT > 0 && first + i < cache.nt) {
                    const hk_f4v a = cache.tri[first + i], b = cache.tri[NT + first + i], c = cache.tri[2 * NT + first + i];
                    T0 = make_float4(a.x, a.y, a.z, a.w), T1 = make_float4(b.x, b.y, b.z, b.w), T2 = make_float4(c.x, c.y, c.z, c.w);
                    asm volatile("" : "+v"(T0.x));   // keeps the arms apart (see NodeCache)
                } else {
                    const float4* tp = sc.leaf_tris + 3 * (size_t)(first + i);
                    T0 = tp[0], T1 = tp[1], T2 = tp[2];
                }
                asm volatile("" ::"v"(T0.x), "v"(T0.y), "v"(T0.z), "v"(T0.w));   // issue the three loads together (see lane_ray_round)
                if (COUNT) ++n_tris;
                float t, u, v;
                if (intersect_triangle(o, d, t_max, mk3(T0.x, T0.y, T0.z), mk3(T1.x, T1.y, T1.z), mk3(T2.x, T2.y, T2.z), t, u, v)) {
                    int prim = __float_as_int(T0.w);
                    if (MODE == 1 && (__float_as_uint(T1.w) & HK_TRI_OPAQUE)) {
                        opaque_hit = true;
                        best.t = t;
                        best.prim = prim;
                        best.u = u;
                        best.v = v;
                        return best;
                    }
                    if (best.prim < 0 || t < best.t || (t == best.t && prim < best.prim)) {
                        best.t = t;
                        best.prim = prim;
                        best.u = u;
                        best.v = v;
                    }
                }
            }
            if (sp > 0) {
                --sp;
                cur = stack[sp * 64 + lane];
            } else
                cur = DONE;
        }
    }
    return best;
}

// ------------------------------------------------------------------------------------------------
// surface geometry at a hit (integrators/volpath/intersection.jl:13-182)
// ------------------------------------------------------------------------------------------------
struct Surface {
    v3 pi, n, ns;
    v2 uv;
    float area;
};
HKD const float* tri_record(const DScene& sc, int prim) { return sc.tri_shade + 32 * (size_t)prim; }   // (DScene::tri_shade != null)
HKD DTriMeta tri_meta(const DScene& sc, int prim) {
    if (sc.tri_shade) {
        const uint32_t* m = reinterpret_cast<const uint32_t*>(tri_record(sc, prim) + 24);
        return DTriMeta{m[0], m[1], m[2]};
    }
    return sc.meta[prim];
}
HKD void tri_vertices(const DScene& sc, int prim, v3& a, v3& b, v3& c) {
    const float* p = sc.positions + 9 * (size_t)prim;
    a = mk3(p[0], p[1], p[2]);
    b = mk3(p[3], p[4], p[5]);
    c = mk3(p[6], p[7], p[8]);
}
HKD v3 geometric_normal(const DScene& sc, int prim) {
    v3 a, b, c;
    tri_vertices(sc, prim, a, b, c);
    return normalize(cross(b - a, c - a));
}
HKD v2 uv_at(const DScene& sc, int prim, float w, float u, float v) {
    if (sc.uvs) {
        const float* q = sc.uvs + 6 * (size_t)prim;
        return mk2(w * q[0] + u * q[2] + v * q[4], w * q[1] + u * q[3] + v * q[5]);
    }
    // default uvs (0,0),(1,0),(1,1)
    return mk2(w * 0.0f + u * 1.0f + v * 1.0f, w * 0.0f + u * 0.0f + v * 1.0f);
}
// Only the fields the BSDFs consume are produced: dpdu/dpdv/shading tangents feed the texture-filter
// context, which texture evaluation ignores (textures/texture-ref.jl:126-135), so they are not computed.
HKD Surface surface_at(const DScene& sc, int prim, float bu, float bv, v3 ro, v3 rd, float t) {
    Surface s;
    float w = 1.0f - bu - bv;
    s.pi = ro + rd * t;
    // (only the shading and light-selection kernels read the packed record: in the traversal kernels the second address path cost
    // k_trace_lean 1.6 ms per Cornell frame without ever being taken, and scenes whose attributes fit in L2 gain nothing — hk_api.cpp
    // builds the records for scenes of >= 32 768 triangles)
    v3 a, b, c;
    if (sc.tri_shade) {
        const float* p = tri_record(sc, prim);
        a = mk3(p[0], p[1], p[2]);
        b = mk3(p[3], p[4], p[5]);
        c = mk3(p[6], p[7], p[8]);
        const float* q = p + 18;   // (a mesh without uvs: the record holds uv_at's defaults — the same products and sums)
        s.uv = mk2(w * q[0] + bu * q[2] + bv * q[4], w * q[1] + bu * q[3] + bv * q[5]);
    } else {
        tri_vertices(sc, prim, a, b, c);
        s.uv = uv_at(sc, prim, w, bu, bv);
    }
    v3 cr = cross(b - a, c - a);
    v3 n = normalize(cr);
    s.area = 0.5f * norm(cr);
    v3 ns = n;
    if (sc.tri_shade || sc.normals) {   // (a packed record of a mesh without vertex normals holds NaNs)
        // the nine floats in three loads up front: read field by field behind the NaN test they become twelve per-lane loads
        typedef float hk_f4u __attribute__((ext_vector_type(4), aligned(4)));
        const float* q = sc.tri_shade ? tri_record(sc, prim) + 9 : sc.normals + 9 * (size_t)prim;
        const hk_f4u q0 = *reinterpret_cast<const hk_f4u*>(q), q1 = *reinterpret_cast<const hk_f4u*>(q + 4);
        const float q8 = q[8];
        if (!(isnan(q0.x) || isnan(q0.w) || isnan(q1.z)))
            ns = normalize(mk3(w * q0.x + bu * q0.w + bv * q1.z, w * q0.y + bu * q1.x + bv * q1.w, w * q0.z + bu * q1.y + bv * q8));
    }
    s.n = dot(n, ns) < 0.0f ? -n : n;
    s.ns = ns;
    return s;
}

// ------------------------------------------------------------------------------------------------
// materials: parameter fetch
// ------------------------------------------------------------------------------------------------
enum { UPLIFT_BOUNDED = 0, UPLIFT_UNBOUNDED = 1 };
template <bool TWO_PLANES = false, bool NO_TEX = false>
HKD float4 rgb_param_coef(const DScene& sc, const DTables& T, const DSpectrumParam& p, const TexCtx& uv, int mode, bool clamp_lo) {
    if (NO_TEX || p.tex < 0) return p.coef;
    float o[4] = {0, 0, 0, 1};
    tex_bilinear(sc.textures[p.tex], uv, o);
    if (clamp_lo) {  // clamp(kd_rgb): [0, Inf)
        o[0] = clampf(o[0], 0.0f, INF_F);
        o[1] = clampf(o[1], 0.0f, INF_F);
        o[2] = clampf(o[2], 0.0f, INF_F);
    }
    return mode == UPLIFT_BOUNDED ? coef_bounded<TWO_PLANES>(T, o[0], o[1], o[2]) : coef_unbounded<TWO_PLANES>(T, o[0], o[1], o[2]);
}
// alpha goes through eval_tex(ctx, ref, uv::Point2f) == _sample_texture_data: NEAREST texel by truncation
// (textures/basic.jl:19-26; spectral-eval.jl:3882-3885), unlike shading, which is bilinear (quirk Q28)
HKD float rgb_param_alpha(const DScene& sc, const DSpectrumParam& p, const TexCtx& uv) {
    if (p.tex < 0) return p.rgba[3];
    const DTexture& t = sc.textures[p.tex];
    if (t.channels < 4 || t.pad == 1) return 1.0f;  // vertex colours: the Point2f method returns RGBSpectrum(0.5f0), alpha 1
    int i = clampi((int)(1.0f + (float)(t.height - 1) * (1.0f - uv.uv.y)), 1, t.height);
    int j = clampi((int)(1.0f + (float)(t.width - 1) * uv.uv.x), 1, t.width);
    return t.data[((size_t)(i - 1) + (size_t)t.height * (size_t)(j - 1)) * 4 + 3];
}
HKD float surface_alpha(const DScene& sc, int mat, const TexCtx& uv) {  // spectral-eval.jl:3882-3888
    const DMaterial& m = sc.materials[mat];
    if (m.kind == HK_MAT_MATTE) return rgb_param_alpha(sc, m.rgb[0], uv);
    return 1.0f;
}
HKD float mix_hash_float(v3 p, v3 wo, const uint32_t* key) {  // mix-material.jl:96-127
    uint64_t h = 0;
    h ^= (uint64_t)__float_as_uint(p.x);
    h *= 0xcc9e2d51ull;
    h ^= (uint64_t)(uint32_t)(__float_as_uint(p.y) << 4);
    h *= 0x1b873593ull;
    h ^= (uint64_t)(uint32_t)(__float_as_uint(p.z) << 8);
    h ^= (uint64_t)(uint32_t)(__float_as_uint(wo.x) << 16);
    h *= 0xcc9e2d51ull;
    h ^= (uint64_t)__float_as_uint(wo.y);
    h *= 0x1b873593ull;
    h ^= (uint64_t)(uint32_t)(__float_as_uint(wo.z) << 12);
    h ^= (uint64_t)key[0] << 24;
    h ^= (uint64_t)key[1];
    h *= 0xcc9e2d51ull;
    h ^= (uint64_t)key[2] << 28;
    h ^= (uint64_t)key[3] << 4;
    h *= 0x1b873593ull;
    h ^= h >> 31;
    h *= 0x7fb5d329728ea185ull;
    h ^= h >> 27;
    h *= 0x81dadef4bc2dd44dull;
    h ^= h >> 33;
    return (float)(uint32_t)(h & 0xFFFFFFFFull) * 2.3283064365386963e-10f;
}
HKD int resolve_mix_material(const DScene& sc, int idx, v3 p, v3 wo, v2 uv) {  // mix-material.jl:222-238
    int cur = idx;
    for (int it = 0; it < 8; ++it) {
        const DMaterial& m = sc.materials[cur];
        if (m.kind != HK_MAT_MIX) return cur;
        float amt = eval_f32_nearest(sc, m, 0, uv);  // eval_tex(ctx, mix.amount, uv::Point2f): nearest texel (Q28)
        if (amt <= 0.0f)
            cur = m.i[0];
        else if (amt >= 1.0f)
            cur = m.i[1];
        else
            cur = amt < mix_hash_float(p, wo, m.mix_key) ? m.i[0] : m.i[1];
    }
    return cur;
}

// ------------------------------------------------------------------------------------------------
// BSDFs (materials/spectral-eval.jl; helpers :3514-3864; reflection/bxdf.jl:67-100; microfacet.jl:83-99)
// ------------------------------------------------------------------------------------------------
struct BSDFSample {
    v3 wi;
    S4 f;
    float pdf;
    bool is_specular;
    float eta_scale;
};
HKD BSDFSample invalid_sample() { return BSDFSample{mk3(0, 0, 1), s4(0.0f), 0.0f, false, 1.0f}; }
HKD void coordinate_system(v3 n, v3& t, v3& b) {
    if (fabsf(n.x) > fabsf(n.y)) {
        float il = 1.0f / sqrtf(n.x * n.x + n.z * n.z);
        t = mk3(n.z * il, 0.0f, -n.x * il);
    } else {
        float il = 1.0f / sqrtf(n.y * n.y + n.z * n.z);
        t = mk3(0.0f, n.z * il, -n.y * il);
    }
    b = cross(n, t);
}
HKD v3 local_to_world(v3 l, v3 n, v3 t, v3 b) { return t * l.x + b * l.y + n * l.z; }
HKD v3 world_to_local(v3 v, v3 n, v3 t, v3 b) { return mk3(dot(v, t), dot(v, b), dot(v, n)); }
HKD v3 reflect(v3 wo, v3 n) { return -wo + 2.0f * dot(wo, n) * n; }
HKD float fresnel_dielectric(float ci, float eta) {
    ci = clampf(ci, -1.0f, 1.0f);
    if (ci < 0.0f) {
        eta = 1.0f / eta;
        ci = -ci;
    }
    float s2i = 1.0f - ci * ci;
    float s2t = s2i / (eta * eta);
    if (s2t >= 1.0f) return 1.0f;
    float ct = sqrtf(1.0f - s2t);
    float rp = (eta * ci - ct) / (eta * ci + ct);
    float rs = (ci - eta * ct) / (ci + eta * ct);
    return 0.5f * (rp * rp + rs * rs);
}
HKD float fr_complex(float ci, float eta, float k) {
    ci = clampf(ci, 0.0f, 1.0f);
    float s2i = 1.0f - ci * ci;
    float eta2 = eta * eta, k2 = k * k;
    float er = eta2 - k2, ei = 2.0f * eta * k;
    float den = er * er + ei * ei;
    float s2tr = s2i * er / den, s2ti = -s2i * ei / den;
    float c2tr = 1.0f - s2tr, c2ti = -s2ti;
    float mag = sqrtf(c2tr * c2tr + c2ti * c2ti);
    float ctr = sqrtf(0.5f * (mag + c2tr));
    float cti = c2ti / (2.0f * ctr);
    if (ctr == 0.0f) cti = sqrtf(0.5f * mag);
    float ecr = eta * ci, eci = k * ci;
    float npr = ecr - ctr, npi = eci - cti, dpr = ecr + ctr, dpi = eci + cti;
    float dpm = dpr * dpr + dpi * dpi;
    float rpr = (npr * dpr + npi * dpi) / dpm, rpi = (npi * dpr - npr * dpi) / dpm;
    float etr = eta * ctr - k * cti, eti = eta * cti + k * ctr;
    float nsr = ci - etr, nsi = -eti, dsr = ci + etr, dsi = eti;
    float dsm = dsr * dsr + dsi * dsi;
    float rsr = (nsr * dsr + nsi * dsi) / dsm, rsi = (nsi * dsr - nsr * dsi) / dsm;
    return ((rpr * rpr + rpi * rpi) + (rsr * rsr + rsi * rsi)) * 0.5f;
}
HKD S4 fr_complex4(float c, S4 eta, S4 k) { return s4(fr_complex(c, eta.x, k.x), fr_complex(c, eta.y, k.y), fr_complex(c, eta.z, k.z), fr_complex(c, eta.w, k.w)); }
HKD float cos2_theta(v3 w) { return w.z * w.z; }
HKD float sin2_theta(v3 w) { return maxf(0.0f, 1.0f - cos2_theta(w)); }
HKD float tan2_theta(v3 w) { return sin2_theta(w) / cos2_theta(w); }
HKD float cos_phi(v3 w) {
    float s = sqrtf(sin2_theta(w));
    return s == 0.0f ? 1.0f : clampf(w.x / s, -1.0f, 1.0f);
}
HKD float sin_phi(v3 w) {
    float s = sqrtf(sin2_theta(w));
    return s == 0.0f ? 0.0f : clampf(w.y / s, -1.0f, 1.0f);
}
HKD bool tr_smooth(float ax, float ay) { return maxf(ax, ay) < 1e-3f; }
HKD float tr_d(v3 wm, float ax, float ay) {
    float t2 = tan2_theta(wm);
    if (isinf(t2)) return 0.0f;
    float c4 = cos2_theta(wm) * cos2_theta(wm);
    if (c4 < 1e-16f) return 0.0f;
    float a = cos_phi(wm) / ax, b = sin_phi(wm) / ay;
    float e = t2 * (a * a + b * b);
    return 1.0f / (PI_F * ax * ay * c4 * ((1.0f + e) * (1.0f + e)));
}
HKD float tr_lambda(v3 w, float ax, float ay) {
    float t2 = tan2_theta(w);
    if (isinf(t2)) return 0.0f;
    float a = cos_phi(w) * ax, b = sin_phi(w) * ay;
    return (sqrtf(1.0f + (a * a + b * b) * t2) - 1.0f) * 0.5f;
}
HKD float tr_g1(v3 w, float ax, float ay) { return 1.0f / (1.0f + tr_lambda(w, ax, ay)); }
HKD float tr_g(v3 wo, v3 wi, float ax, float ay) { return 1.0f / (1.0f + tr_lambda(wo, ax, ay) + tr_lambda(wi, ax, ay)); }
HKD float tr_pdf(v3 w, v3 wm, float ax, float ay) { return tr_g1(w, ax, ay) / fabsf(w.z) * tr_d(wm, ax, ay) * fabsf(dot(w, wm)); }
HKD v3 tr_sample_wm(v3 w, v2 u, float ax, float ay) {
    v3 wh = normalize(mk3(ax * w.x, ay * w.y, w.z));
    if (wh.z < 0.0f) wh = -wh;
    v3 t1 = wh.z < 0.99999f ? normalize(cross(mk3(0, 0, 1), wh)) : mk3(1, 0, 0);
    v3 t2 = cross(wh, t1);
    float r = sqrtf(u.x);
    float phi = 2.0f * PI_F * u.y;
    float sphi, cphi;
    jl_sincos(phi, sphi, cphi);
    float px = r * cphi, py = r * sphi;
    float h = sqrtf(1.0f - px * px);
    py = lerpf(h, py, 0.5f * (1.0f + wh.z));
    float pz = sqrtf(maxf(0.0f, 1.0f - px * px - py * py));
    v3 nh = px * t1 + py * t2 + pz * wh;
    return normalize(mk3(ax * nh.x, ay * nh.y, maxf(1e-6f, nh.z)));
}
HKD float pl_sample(const DPLSpectrum& s, float l) {  // spectral/piecewise-linear.jl
    int N = s.n;
    if (l <= s.lambdas[0]) return s.values[0];
    if (l >= s.lambdas[N - 1]) return s.values[N - 1];
    int lo = 1, hi = N;
    while (lo + 1 < hi) {
        int mid = (lo + hi) >> 1;
        if (s.lambdas[mid - 1] <= l)
            lo = mid;
        else
            hi = mid;
    }
    float t = (l - s.lambdas[lo - 1]) / (s.lambdas[hi - 1] - s.lambdas[lo - 1]);
    return s.values[lo - 1] * (1.0f - t) + s.values[hi - 1] * t;
}
HKD S4 eval_ior(const DScene& sc, const DTables& T, const DMaterial& m, int slot, const TexCtx& uv, S4 lambda) {
    if (m.spectrum[slot] >= 0) {
        const DPLSpectrum& s = sc.spectra[m.spectrum[slot]];
        return s4(pl_sample(s, lambda.x), pl_sample(s, lambda.y), pl_sample(s, lambda.z), pl_sample(s, lambda.w));
    }
    return eval_scaled(rgb_param_coef(sc, T, m.rgb[slot], uv, UPLIFT_UNBOUNDED, false), lambda);
}
HKD BSDFSample sample_lambert(v3 wo, v3 n, v2 u, S4 f) {
    float wdn = dot(wo, n);
    v3 t, b;
    coordinate_system(n, t, b);
    v3 lw = cosine_sample_hemisphere(u);
    float ct = lw.z;
    if (ct < 1e-6f) return invalid_sample();
    if (wdn < 0.0f) lw = mk3(lw.x, lw.y, -lw.z);
    BSDFSample s;
    s.wi = normalize(local_to_world(lw, n, t, b));
    s.f = f;
    s.pdf = ct / PI_F;
    s.is_specular = false;
    s.eta_scale = 1.0f;
    return s;
}

}  // namespace hkd
#include "hk_layered.h"
namespace hkd {

// MatteMaterial with its reflectance spectrum already evaluated: k_shade evaluates kd = uplift(clamp(Kd))(lambda) ONCE per vertex and
// hands it to both the next-event evaluation and the BSDF sample (the two would each run the four sigmoid evaluations — a
// square root and a division per wavelength — on the same inputs).  Same operations, same order as the generic entry points.
template <bool NO_TEX = false>
HKD S4 matte_kd(const DScene& sc, const DTables& T, const DMaterial& m, const TexCtx& uv, S4 lambda) {
    return eval_bounded(rgb_param_coef<true, NO_TEX>(sc, T, m.rgb[0], uv, UPLIFT_BOUNDED, true), lambda);
}
HKD S4 eval_matte_kd(S4 kd, v3 wo_w, v3 wi_w, v3 n, float& pdf) {  // spectral-eval.jl:371-398
    pdf = 0.0f;
    float ci = dot(wi_w, n), co = dot(wo_w, n);
    if (ci * co < 0.0f) return s4(0.0f);
    float ct = fabsf(ci);
    if (ct < 1e-6f) return s4(0.0f);
    pdf = ct / PI_F;
    return kd / PI_F;
}
HKD BSDFSample sample_lambert(v3 wo, v3 n, v2 u, S4 f);
template <bool NO_TEX = false>
HKD BSDFSample sample_matte_kd(const DScene& sc, const DMaterial& m, S4 kd, v3 wo_w, v3 n, const TexCtx& uv, v2 u) {  // :42-101
    float wdn = dot(wo_w, n);
    if (fabsf(wdn) < 1e-6f) return invalid_sample();
    float sigma = eval_f32<NO_TEX>(sc, m, 0, uv);
    S4 f;
    if (sigma > 0.0f) {
        float rf = 1.0f - 0.5f * sigma / (sigma + 0.33f);
        f = kd * (rf / PI_F);
    } else
        f = kd * (1.0f / PI_F);
    return sample_lambert(wo_w, n, u, f);
}

// KIND is a compile-time constant in the per-kind shade kernels (material-sorted queues)
template <int KIND>
HKD BSDFSample sample_bsdf(const DScene& sc, const DTables& T, const DMaterial& m, v3 wo_w, v3 n, const TexCtx& uv, S4 lambda, v2 u, float uc, bool regularize) {
    if (KIND == HK_MAT_MATTE) {  // spectral-eval.jl:42-101
        float wdn = dot(wo_w, n);
        if (fabsf(wdn) < 1e-6f) return invalid_sample();
        S4 kd = eval_bounded(rgb_param_coef(sc, T, m.rgb[0], uv, UPLIFT_BOUNDED, true), lambda);
        float sigma = eval_f32(sc, m, 0, uv);
        S4 f;
        if (sigma > 0.0f) {
            float rf = 1.0f - 0.5f * sigma / (sigma + 0.33f);
            f = kd * (rf / PI_F);
        } else
            f = kd * (1.0f / PI_F);
        return sample_lambert(wo_w, n, u, f);
    } else if (KIND == HK_MAT_MIRROR) {  // :108-132
        float wdn = dot(wo_w, n);
        if (fabsf(wdn) < 1e-6f) return invalid_sample();
        BSDFSample s;
        s.wi = reflect(wo_w, wdn < 0.0f ? -n : n);
        s.f = eval_bounded(rgb_param_coef(sc, T, m.rgb[0], uv, UPLIFT_BOUNDED, false), lambda);
        s.pdf = 1.0f;
        s.is_specular = true;
        s.eta_scale = 1.0f;
        return s;
    } else if (KIND == HK_MAT_GLASS) {  // :140-198
        float ior = eval_f32(sc, m, 0, uv);
        if (ior == 0.0f) ior = 1.0f;
        float co = dot(wo_w, n);
        bool entering = co > 0.0f;
        v3 no = entering ? n : -n;
        co = fabsf(co);
        float eta = entering ? ior : (1.0f / ior);
        float F = fresnel_dielectric(co, eta);
        BSDFSample s;
        s.pdf = 1.0f;
        s.is_specular = true;
        s.eta_scale = 1.0f;
        bool do_reflect = uc < F;
        float ctt = 0.0f;
        if (!do_reflect) {
            float s2i = maxf(0.0f, 1.0f - co * co);
            float s2t = s2i / (eta * eta);
            if (s2t >= 1.0f)
                do_reflect = true;
            else
                ctt = sqrtf(1.0f - s2t);
        }
        if (do_reflect) {
            s.wi = reflect(wo_w, no);
            s.f = eval_bounded(rgb_param_coef(sc, T, m.rgb[0], uv, UPLIFT_BOUNDED, false), lambda);
            return s;
        }
        s.wi = normalize(-wo_w / eta + (co / eta - ctt) * no);
        s.f = eval_bounded(rgb_param_coef(sc, T, m.rgb[1], uv, UPLIFT_BOUNDED, false), lambda);
        s.eta_scale = 1.0f / (eta * eta);
        return s;
    } else if (KIND == HK_MAT_CONDUCTOR) {  // :223-318
        v3 t, b;
        coordinate_system(n, t, b);
        v3 wo = world_to_local(wo_w, n, t, b);
        if (wo.z == 0.0f) return invalid_sample();
        float rough = eval_f32(sc, m, 0, uv);
        float ax = (m.flags & HK_MATF_REMAP_ROUGHNESS) ? sqrtf(rough) : rough;
        float ay = ax;
        if (regularize) {
            ax = ax < 0.3f ? clampf(2.0f * ax, 0.1f, 0.3f) : ax;
            ay = ay < 0.3f ? clampf(2.0f * ay, 0.1f, 0.3f) : ay;
        }
        if (!tr_smooth(ax, ay)) {
            ax = maxf(ax, 1e-4f);
            ay = maxf(ay, 1e-4f);
        }
        S4 eta = eval_ior(sc, T, m, 0, uv, lambda), k = eval_ior(sc, T, m, 1, uv, lambda);
        BSDFSample s;
        s.eta_scale = 1.0f;
        if (tr_smooth(ax, ay)) {
            v3 wi = mk3(-wo.x, -wo.y, wo.z);
            float ci = fabsf(wi.z);
            s.f = fr_complex4(ci, eta, k) / ci;
            s.wi = local_to_world(wi, n, t, b);
            s.pdf = 1.0f;
            s.is_specular = true;
            return s;
        }
        v3 wm = tr_sample_wm(wo, u, ax, ay);
        v3 wi = -wo + 2.0f * dot(wo, wm) * wm;
        if (!(wo.z * wi.z > 0.0f)) return invalid_sample();
        float pdf = tr_pdf(wo, wm, ax, ay) / (4.0f * fabsf(dot(wo, wm)));
        float co = fabsf(wo.z), ci = fabsf(wi.z);
        if (ci == 0.0f || co == 0.0f) return invalid_sample();
        S4 F = fr_complex4(fabsf(dot(wo, wm)), eta, k);
        float D = tr_d(wm, ax, ay), G = tr_g(wo, wi, ax, ay);
        s.f = D * F * G / (4.0f * ci * co);
        s.wi = local_to_world(wi, n, t, b);
        s.pdf = pdf;
        s.is_specular = false;
        return s;
    } else if (KIND == HK_MAT_COATED_DIFFUSE) {  // :1233-1441
        return layered_sample<false>(layered_params<false>(sc, T, m, uv, lambda, regularize), wo_w, n, u, uc);
    } else if (KIND == HK_MAT_COATED_DIFFUSE_TRANSMISSION) {  // :2341-2497
        return layered_sample<true>(layered_params<true>(sc, T, m, uv, lambda, regularize), wo_w, n, u, uc);
    } else if (KIND == HK_MAT_COATED_CONDUCTOR) {  // :2877-3231
        return cc_sample(cc_params(sc, T, m, uv, lambda, regularize), wo_w, n, u, uc);
    } else if (KIND == HK_MAT_THIN_DIELECTRIC) {  // :1975-2037
        return thin_dielectric_sample(eval_f32(sc, m, 0, uv), wo_w, n, uc);
    } else if (KIND == HK_MAT_DIFFUSE_TRANSMISSION) {  // :2083-2163
        return dt_sample(dt_params(sc, T, m, uv, lambda), wo_w, n, u, uc);
    } else {  // generic fallback (:322-359): gray 0.5 Lambertian (Q24)
        float wdn = dot(wo_w, n);
        if (fabsf(wdn) < 1e-6f) return invalid_sample();
        return sample_lambert(wo_w, n, u, s4(0.5f) * (1.0f / PI_F));
    }
}
template <int KIND>
HKD S4 eval_bsdf(const DScene& sc, const DTables& T, const DMaterial& m, v3 wo_w, v3 wi_w, v3 n, const TexCtx& uv, S4 lambda, float& pdf) {
    pdf = 0.0f;
    if (KIND == HK_MAT_MATTE) {  // :371-398
        float ci = dot(wi_w, n), co = dot(wo_w, n);
        if (ci * co < 0.0f) return s4(0.0f);
        float ct = fabsf(ci);
        if (ct < 1e-6f) return s4(0.0f);
        S4 kd = eval_bounded(rgb_param_coef(sc, T, m.rgb[0], uv, UPLIFT_BOUNDED, true), lambda);
        pdf = ct / PI_F;
        return kd / PI_F;
    } else if (KIND == HK_MAT_MIRROR || KIND == HK_MAT_GLASS) {
        return s4(0.0f);
    } else if (KIND == HK_MAT_CONDUCTOR) {  // :422-488
        v3 t, b;
        coordinate_system(n, t, b);
        v3 wo = world_to_local(wo_w, n, t, b), wi = world_to_local(wi_w, n, t, b);
        if (!(wo.z * wi.z > 0.0f)) return s4(0.0f);
        float rough = eval_f32(sc, m, 0, uv);
        float ax = (m.flags & HK_MATF_REMAP_ROUGHNESS) ? sqrtf(rough) : rough;
        float ay = ax;
        if (!tr_smooth(ax, ay)) {
            ax = maxf(ax, 1e-4f);
            ay = maxf(ay, 1e-4f);
        }
        if (tr_smooth(ax, ay)) return s4(0.0f);
        float co = fabsf(wo.z), ci = fabsf(wi.z);
        if (ci == 0.0f || co == 0.0f) return s4(0.0f);
        v3 wm = wi + wo;
        if (dot(wm, wm) == 0.0f) return s4(0.0f);
        wm = normalize(wm);
        S4 eta = eval_ior(sc, T, m, 0, uv, lambda), k = eval_ior(sc, T, m, 1, uv, lambda);
        S4 F = fr_complex4(fabsf(dot(wo, wm)), eta, k);
        float D = tr_d(wm, ax, ay), G = tr_g(wo, wi, ax, ay);
        S4 f = D * F * G / (4.0f * ci * co);
        v3 wmp = wm.z < 0.0f ? -wm : wm;
        pdf = tr_pdf(wo, wmp, ax, ay) / (4.0f * fabsf(dot(wo, wmp)));
        return f;
    } else if (KIND == HK_MAT_COATED_DIFFUSE) {  // :1563-1937
        return layered_eval<false>(layered_params<false>(sc, T, m, uv, lambda, false), wo_w, wi_w, n, pdf);
    } else if (KIND == HK_MAT_COATED_DIFFUSE_TRANSMISSION) {  // :2501-2832
        return layered_eval<true>(layered_params<true>(sc, T, m, uv, lambda, false), wo_w, wi_w, n, pdf);
    } else if (KIND == HK_MAT_COATED_CONDUCTOR) {  // :3238-3420
        return cc_eval(cc_params(sc, T, m, uv, lambda, false), wo_w, wi_w, n, pdf);
    } else if (KIND == HK_MAT_THIN_DIELECTRIC) {  // :2045-2051
        return s4(0.0f);
    } else if (KIND == HK_MAT_DIFFUSE_TRANSMISSION) {  // :2170-2218
        return dt_eval(dt_params(sc, T, m, uv, lambda), wo_w, wi_w, n, pdf);
    } else {  // fallback (:491-511)
        float ci = dot(wi_w, n), co = dot(wo_w, n);
        if (ci * co < 0.0f) return s4(0.0f);
        float ct = fabsf(ci);
        if (ct < 1e-6f) return s4(0.0f);
        pdf = ct / PI_F;
        return s4(0.5f / PI_F);
    }
}

// ------------------------------------------------------------------------------------------------
// environment map (textures/environment_map.jl:78-160, 200-229, 290-371; sampler/sampling.jl:264-361)
// ------------------------------------------------------------------------------------------------
HKD v2 equal_area_sphere_to_square(v3 d) {
    float x = fabsf(d.x), y = fabsf(d.y), z = fabsf(d.z);
    // quirk Q35: the reference takes sqrt(1 - |z|) bare (environment_map.jl:82; pbrt-v4 has SafeSqrt there) and a NORMALISED direction can
    // carry |z| = 1 + 2^-23: a DomainError on its CPU path, NaN -> an undefined texel index on a GPU backend.  Where the reference is defined
    // (|z| <= 1) max(0, .) changes nothing; beyond, the pole's texel is read (what pbrt does) instead of a NaN that would stay in the pixel.
    float r = sqrtf(maxf(0.0f, 1.0f - z));
    float a = maxf(x, y);
    float b = a == 0.0f ? 0.0f : minf(x, y) / a;
    const float t1 = 0.406758566246788489601959989e-5f, t2 = 0.636226545274016134946890922156f, t3 = 0.61572017898280213493197203466e-2f,
                t4 = -0.247333733281268944196501420480f, t5 = 0.881770664775316294736387951347e-1f, t6 = 0.419038818029165735901852432784e-1f,
                t7 = -0.251390972343483509333252996350e-1f;
    float phi = t1 + b * (t2 + b * (t3 + b * (t4 + b * (t5 + b * (t6 + b * t7)))));
    if (x < y) phi = 1.0f - phi;
    float v = phi * r;
    float u = r - v;
    if (d.z < 0.0f) {
        float tu = u;
        u = 1.0f - v;
        v = 1.0f - tu;
    }
    u = copysignf(u, d.x);
    v = copysignf(v, d.y);
    return mk2(0.5f * (u + 1.0f), 0.5f * (v + 1.0f));
}
HKD v3 equal_area_square_to_sphere(v2 p) {
    float u = 2.0f * p.x - 1.0f, v = 2.0f * p.y - 1.0f;
    float up = fabsf(u), vp = fabsf(v);
    float sd = 1.0f - (up + vp);
    float r = 1.0f - fabsf(sd);
    float phi = (r == 0.0f ? 1.0f : (vp - up) / r + 1.0f) * PI_F / 4.0f;
    float z = copysignf(1.0f - r * r, sd);
    float sphi, cphi;
    jl_sincos(phi, sphi, cphi);
    float cp = copysignf(cphi, u), sp = copysignf(sphi, v);
    float rc = r * sqrtf(2.0f - r * r);
    return mk3(cp * rc, sp * rc, z);
}
HKD v2 env_direction_to_uv(const DEnvMap& e, v3 d) {  // transpose(rotation) * dir
    const float* R = e.rot;
    return equal_area_sphere_to_square(mk3(R[0] * d.x + R[3] * d.y + R[6] * d.z, R[1] * d.x + R[4] * d.y + R[7] * d.z, R[2] * d.x + R[5] * d.y + R[8] * d.z));
}
HKD v3 env_uv_to_direction(const DEnvMap& e, v2 uv) {  // rotation * dir_light
    const float* R = e.rot;
    v3 d = equal_area_square_to_sphere(uv);
    return mk3(R[0] * d.x + R[1] * d.y + R[2] * d.z, R[3] * d.x + R[4] * d.y + R[5] * d.z, R[6] * d.x + R[7] * d.y + R[8] * d.z);
}
HKD float4 env_texel(const DEnvMap& e, int y1, int x1) { return e.data[(size_t)(y1 - 1) + (size_t)e.height * (size_t)(x1 - 1)]; }
HKD float4 env_eval(const DEnvMap& e, v3 dir) {  // bilinear: escaped rays
    v2 uv = env_direction_to_uv(e, dir);
    int h = e.height, w = e.width;
    float x = uv.x * (float)(w - 1) + 1.0f, y = uv.y * (float)(h - 1) + 1.0f;
    float flx = floorf(x), fly = floorf(y);
    int x0 = (int)flx, y0 = (int)fly;
    int x1 = x0 + 1, y1 = y0 + 1;
    x0 = clampi(x0, 1, w), x1 = clampi(x1, 1, w), y0 = clampi(y0, 1, h), y1 = clampi(y1, 1, h);
    x1 = x1 > w ? 1 : x1;
    float fx = x - flx, fy = y - fly;
    float4 c00 = env_texel(e, y0, x0), c10 = env_texel(e, y0, x1), c01 = env_texel(e, y1, x0), c11 = env_texel(e, y1, x1);
    float4 o;
    o.x = (c00.x * (1.0f - fx) + c10.x * fx) * (1.0f - fy) + (c01.x * (1.0f - fx) + c11.x * fx) * fy;
    o.y = (c00.y * (1.0f - fx) + c10.y * fx) * (1.0f - fy) + (c01.y * (1.0f - fx) + c11.y * fx) * fy;
    o.z = (c00.z * (1.0f - fx) + c10.z * fx) * (1.0f - fy) + (c01.z * (1.0f - fx) + c11.z * fx) * fy;
    o.w = 1.0f;
    return o;
}
HKD float4 env_lookup_uv(const DEnvMap& e, v2 uv) {  // nearest: sampled directions
    int ui = clampi((int)floorf(uv.x * (float)e.width) + 1, 1, e.width);
    int vi = clampi((int)floorf(uv.y * (float)e.height) + 1, 1, e.height);
    return env_texel(e, vi, ui);
}
HKD int find_interval_binary20(const float* cdf, int n, float u) {  // sampling.jl:305-333
    int lo = 1, hi = n;
    // hi - lo at least halves per step; once lo == hi the remaining steps of the reference's fixed 20 re-test cdf[lo - 1] <= u
    // (true by construction) and leave lo alone: floor(log2(n - 1)) + 1 steps give the identical index (10 for a 512-row map)
    const int steps = n <= 1 ? 0 : (32 - __builtin_clz((unsigned)(n - 1)));
    for (int k = 0; k < steps && k < 20; ++k) {
        int mid = (lo + hi + 1) / 2;
        bool c = cdf[mid - 1] <= u;
        lo = c ? mid : lo;
        hi = c ? hi : mid - 1;
    }
    return lo;
}
HKD v2 dist2d_sample(const DEnvMap& e, v2 u, float& pdf) {
    const int nu = e.nu, nv = e.nv;
    int vo = clampi(find_interval_binary20(e.marg_cdf, nv + 1, u.y), 1, nv);
    float c0 = e.marg_cdf[vo - 1], c1 = e.marg_cdf[vo];
    float du_v = u.y - c0, den_v = c1 - c0;
    if (den_v > 0.0f) du_v /= den_v;
    float v_s = ((float)(vo - 1) + du_v) / (float)nv;
    float pdf_v = e.marg_func_int > 0.0f ? e.marg_func[vo - 1] / e.marg_func_int : 0.0f;
    const float* ccdf = e.cond_cdf + (size_t)(vo - 1) * (size_t)(nu + 1);
    int uo = clampi(find_interval_binary20(ccdf, nu + 1, u.x), 1, nu);
    float d0 = ccdf[uo - 1], d1 = ccdf[uo];
    float du_u = u.x - d0, den_u = d1 - d0;
    if (den_u > 0.0f) du_u /= den_u;
    float u_s = ((float)(uo - 1) + du_u) / (float)nu;
    float fiv = e.cond_func_int[vo - 1];
    float pdf_u = fiv > 0.0f ? e.cond_func[(size_t)(vo - 1) * nu + (uo - 1)] / fiv : 0.0f;
    pdf = pdf_u * pdf_v;
    return mk2(u_s, v_s);
}
HKD float env_pdf_li(const DEnvMap& e, v3 wi) {  // lights.jl:336-347
    v2 uv = env_direction_to_uv(e, wi);
    int iu = clampi((int)floorf(uv.x * (float)e.nu) + 1, 1, e.nu);
    int iv = clampi((int)floorf(uv.y * (float)e.nv) + 1, 1, e.nv);
    return (e.cond_func[(size_t)(iv - 1) * e.nu + (iu - 1)] / e.marg_func_int) / (4.0f * PI_F);
}

// ------------------------------------------------------------------------------------------------
// lights (integrators/physical-wavefront/lights.jl:39-297; lights/diffuse-area.jl:53-82)
// ------------------------------------------------------------------------------------------------
struct LightSample {
    S4 Li;
    v3 wi;
    float pdf;
    v3 p_light;
    bool is_delta;
};
HKD S4 light_spectrum(const DLight& l, S4 lambda) { return eval_illuminant(l.coef, lambda); }
// TWO_PLANES: see rgb_to_spectrum (true in k_shade<Matte>).  SIMPLE: the scene has no ambient / environment light and no texture
// (DScene::simple_lights): those branches are compiled out — k_shade<Matte> then spills 29 registers instead of 73 (Cornell -10 %)
template <bool TWO_PLANES = false, bool SIMPLE = false>
HKD S4 arealight_Le(const DScene& sc, const DTables& T, const DLight& l, v3 wo, v3 n, v2 uv, S4 lambda) {
    if (l.kind != HK_LIGHT_DIFFUSE_AREA) return s4(0.0f);
    if (!(l.flags & 1) && dot(wo, n) < 0.0f) return s4(0.0f);
    if (l.Le_tex < 0) return eval_bounded(l.coef, lambda);  // uplift_rgb(Le * scale), bounded (Q3)
    if (SIMPLE) return s4(0.0f);   // no textured emitters in such a scene
    float o[4] = {0, 0, 0, 1};
    tex_bilinear(sc.textures[l.Le_tex], uv, o);
    return eval_bounded(coef_bounded<TWO_PLANES>(T, o[0] * l.scale, o[1] * l.scale, o[2] * l.scale), lambda);
}
template <bool TWO_PLANES = false, bool SIMPLE = false>
HKD LightSample sample_light(const DScene& sc, const DTables& T, const DLight& l, v3 p, S4 lambda, v2 u) {
    LightSample s;
    s.Li = s4(0.0f);
    s.wi = mk3(0, 0, 1);
    s.pdf = 0.0f;
    s.p_light = mk3(0, 0, 0);
    s.is_delta = false;
    switch (l.kind) {
        case HK_LIGHT_POINT:
        case HK_LIGHT_SPOT: {
            v3 pos = mk3(l.p[0], l.p[1], l.p[2]);
            v3 tl = pos - p;
            float d2 = dot(tl, tl);
            float dist = sqrtf(d2);
            if (dist < 1e-6f) return s;
            v3 wi = tl / dist;
            float falloff = 1.0f;
            if (l.kind == HK_LIGHT_SPOT) {
                v3 mw = -wi;
                v3 wl = normalize(mk3(l.v[0] * mw.x + l.v[1] * mw.y + l.v[2] * mw.z, l.v[3] * mw.x + l.v[4] * mw.y + l.v[5] * mw.z,
                                      l.v[6] * mw.x + l.v[7] * mw.y + l.v[8] * mw.z));
                float ct = wl.z;
                if (ct < l.cos_total_width) return s;
                if (!(ct >= l.cos_falloff_start)) {
                    float dl = (ct - l.cos_total_width) / (l.cos_falloff_start - l.cos_total_width);
                    falloff = dl * dl * dl * dl;
                }
                s.Li = ((l.scale * light_spectrum(l, lambda)) * falloff) / d2;
            } else
                s.Li = (l.scale * light_spectrum(l, lambda)) / d2;
            s.wi = wi;
            s.pdf = 1.0f;
            s.p_light = pos;
            s.is_delta = true;
            return s;
        }
        case HK_LIGHT_DIRECTIONAL:
        case HK_LIGHT_SUN: {
            v3 wi = -mk3(l.p[0], l.p[1], l.p[2]);
            s.wi = wi;
            s.p_light = p + 1.0e6f * wi;
            s.Li = l.scale * light_spectrum(l, lambda);
            s.pdf = 1.0f;
            s.is_delta = true;
            return s;
        }
        case HK_LIGHT_AMBIENT: {
            if (SIMPLE) return s;   // compiled out: the scene has none (DScene::simple_lights)
            float z = 1.0f - 2.0f * u.x;
            float r = sqrtf(maxf(0.0f, 1.0f - z * z));
            float phi = 2.0f * PI_F * u.y;
            float sphi, cphi;
            jl_sincos(phi, sphi, cphi);
            v3 wi = mk3(r * cphi, r * sphi, z);
            s.wi = wi;
            s.pdf = 1.0f / (4.0f * PI_F);
            s.p_light = p + 1.0e6f * wi;
            s.Li = l.scale * light_spectrum(l, lambda);
            return s;
        }
        case HK_LIGHT_ENVIRONMENT: {  // lights.jl:158-190
            if (SIMPLE) return s;
            const DEnvMap& e = sc.envmaps[l.Le_tex];
            float map_pdf;
            v2 uv = dist2d_sample(e, u, map_pdf);
            v3 wi = env_uv_to_direction(e, uv);
            float pdf = map_pdf / (4.0f * PI_F);
            if (pdf <= 0.0f) return s;
            float4 t = env_lookup_uv(e, uv);
            s.wi = wi;
            s.pdf = pdf;
            s.p_light = p + 1.0e6f * wi;
            s.Li = eval_illuminant(coef_illuminant<TWO_PLANES>(T, t.x * l.Le_rgba[0], t.y * l.Le_rgba[1], t.z * l.Le_rgba[2]), lambda);
            return s;
        }
        case HK_LIGHT_DIFFUSE_AREA: {
            float b0, b1;
            if (u.x < u.y) {
                b0 = u.x / 2.0f;
                b1 = u.y - b0;
            } else {
                b1 = u.y / 2.0f;
                b0 = u.x - b1;
            }
            float b2 = 1.0f - b0 - b1;
            v3 pl = b0 * mk3(l.v[0], l.v[1], l.v[2]) + b1 * mk3(l.v[3], l.v[4], l.v[5]) + b2 * mk3(l.v[6], l.v[7], l.v[8]);
            v3 tl = pl - p;
            float d2 = dot(tl, tl);
            if (d2 < 1e-12f) return s;
            float dist = sqrtf(d2);
            v3 wi = tl / dist;
            v3 ln = mk3(l.normal[0], l.normal[1], l.normal[2]);
            float ct = fabsf(dot(ln, -wi));
            if (ct < 1e-6f) return s;
            float pdf = d2 / (ct * l.area);
            v2 uvs = mk2(b0 * l.uv[0] + b1 * l.uv[2] + b2 * l.uv[4], b0 * l.uv[1] + b1 * l.uv[3] + b2 * l.uv[5]);
            S4 Le = arealight_Le<TWO_PLANES, SIMPLE>(sc, T, l, mk3(-wi.x, -wi.y, -wi.z), ln, uvs, lambda);
            if (is_black(Le)) return s;
            s.Li = Le;
            s.wi = wi;
            s.pdf = pdf;
            s.p_light = pl;
            return s;
        }
        default: return s;
    }
}

// ---- light BVH (lights/bvh-light-sampler.jl:58-232, light-bounds.jl:96-109,177-182) ----
HKD float cos_sub_clamped(float sA, float cA, float sB, float cB) { return cA > cB ? 1.0f : cA * cB + sA * sB; }
HKD float sin_sub_clamped(float sA, float cA, float sB, float cB) { return cA > cB ? 0.0f : sA * cB - cA * sB; }
// sqrtf(x) for x == 0 or 2^-96 <= x < 2^32, bit for bit: the compiler's correctly rounded square root (hardware v_sqrt_f32 within 1 ulp,
// then the neighbour whose residual has the right sign) WITHOUT the range scaling for tiny arguments and the class check for inf / nan
// that the general expansion carries — 11 vector instructions instead of 16 and three of its five hazard s_nops.  The arguments below
// are max(0, 1 - c^2) and 1 - r^2 / d^2 of cosines in [-1, 1]: 0 or at least 2^-24.  Exhaustively equal to sqrtf on the whole domain
// (tools/sqrt_exact.hip; tests/test_gpu_parity.py::test_light_bvh_parity holds the pmf at 0 ulp from the oracle).
HKD float sqrt_unit(float x) {
    float s = __builtin_amdgcn_sqrtf(x);
    const float dn = __uint_as_float(__float_as_uint(s) - 1u), up = __uint_as_float(__float_as_uint(s) + 1u);
    const float e_dn = __builtin_fmaf(-dn, s, x), e_up = __builtin_fmaf(-up, s, x);
    s = 0.0f >= e_dn ? dn : s;
    s = 0.0f < e_up ? up : s;
    return x != x ? x : s;   // (a NaN comes back as it came, as from the general expansion: p on a node's centre)
}
HKD float node_importance(const DLightNode& nd, v3 p, v3 n) {
    if (nd.phi == 0.0f) return 0.0f;
    v3 pc = mk3(nd.centre[0], nd.centre[1], nd.centre[2]);
    v3 dp = p - pc;
    float dd = dot(dp, dp);
    float d2 = maxf(dd, nd.half_diag);
    v3 wi = normalize(dp);
    float cw = dot(mk3(nd.w[0], nd.w[1], nd.w[2]), wi);
    if (nd.bits & 1u) cw = fabsf(cw);
    float sw = sqrt_unit(maxf(0.0f, 1.0f - cw * cw));
    float cb = dd < nd.r2 ? -1.0f : sqrt_unit(maxf(0.0f, 1.0f - nd.r2 / dd));
    float sb = sqrt_unit(maxf(0.0f, 1.0f - cb * cb));
    float so = nd.sin_o;
    float cx = cos_sub_clamped(sw, cw, so, nd.cos_o);
    float sx = sin_sub_clamped(sw, cw, so, nd.cos_o);
    float cp = cos_sub_clamped(sx, cx, sb, cb);
    if (cp <= nd.cos_e) return 0.0f;
    float imp = nd.phi * cp / d2;
    if (!is_zero(n)) {
        float ci = fabsf(dot(wi, n));
        float si = sqrt_unit(maxf(0.0f, 1.0f - ci * ci));
        imp *= cos_sub_clamped(si, ci, sb, cb);
    }
    return maxf(imp, 0.0f);
}
// A node is fetched whole (4 x 16-B loads) into registers before any of its fields is looked at: field-by-field access behind the
// early returns of node_importance compiles to 14 separate dword loads per node, and with every lane on its own node the walk is
// bound by per-lane cache-line lookups (42 per level before, 8 now; 10^6-triangle / 5*10^4-light scene: light sampling was 71 %
// of k_shade).  The walk also keeps the chosen child's (bits, child) pair instead of re-reading the node it just evaluated.
HKD DLightNode load_light_node(const DLightNode* __restrict__ nodes, int idx0) {
    const float4* q = reinterpret_cast<const float4*>(nodes + idx0);
    float4 a = q[0], b = q[1], c = q[2], d = q[3];
    DLightNode n;
    n.centre[0] = a.x, n.centre[1] = a.y, n.centre[2] = a.z, n.half_diag = a.w;
    n.r2 = b.x, n.w[0] = b.y, n.w[1] = b.z, n.w[2] = b.w;
    n.phi = c.x, n.cos_o = c.y, n.cos_e = c.z, n.sin_o = c.w;
    n.bits = __float_as_uint(d.x), n.child1_or_light = __float_as_uint(d.y);
    n.pad[0] = n.pad[1] = 0u;
    return n;
}
HKD int bvh_sample_light(const DScene& sc, v3 p, v3 n, float u, float& pmf_out, unsigned& visited) {
    pmf_out = 0.0f;
    int ninf = sc.num_infinite_lights, nbvh = sc.num_bvh_lights;
    if (ninf + nbvh == 0) return 0;
    bool has_bvh = nbvh > 0;
    float p_inf = (float)ninf / (float)(ninf + (has_bvh ? 1 : 0));
    if (ninf > 0 && u < p_inf) {
        float ur = u / p_inf;
        int idx = (int)floorf(ur * (float)ninf);
        idx = (idx < ninf - 1 ? idx : ninf - 1) + 1;
        pmf_out = p_inf / (float)ninf;
        return sc.infinite_lights[idx - 1];
    }
    if (!has_bvh) return 0;
    float ub = ninf > 0 ? minf((u - p_inf) / (1.0f - p_inf), 0.99999994f) : minf(u, 0.99999994f);
    float pmf = 1.0f - p_inf;
    uint32_t bits, child;
    {
        const DLightNode root = load_light_node(sc.lnodes, 0);
        bits = root.bits, child = root.child1_or_light;
    }
    for (int it = 0; it < 64; ++it) {
        if (bits & 2u) {
            pmf_out = pmf;
            return (int)child;
        }
        const DLightNode n0 = load_light_node(sc.lnodes, (int)child), n1 = load_light_node(sc.lnodes, (int)child + 1);   // the sibling pair: one 128-B line
        float c0 = node_importance(n0, p, n);
        float c1 = node_importance(n1, p, n);
        visited += 2;
        if (c0 == 0.0f && c1 == 0.0f) return 0;
        float p0 = c0 / (c0 + c1);
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
    return 0;
}
HKD float bvh_pmf(const DScene& sc, v3 p, v3 n, int light_1based, unsigned& visited) {
    if (light_1based < 1) return 0.0f;
    bool has_bvh = sc.num_bvh_lights > 0;
    uint32_t trail = sc.bit_trails[light_1based - 1];
    if (trail == 0xFFFFFFFFu) {
        if (sc.num_infinite_lights == 0) return 0.0f;
        return 1.0f / (float)(sc.num_infinite_lights + (has_bvh ? 1 : 0));
    }
    if (!has_bvh) return 0.0f;
    float p_inf = (float)sc.num_infinite_lights / (float)(sc.num_infinite_lights + 1);
    float pm = 1.0f - p_inf;
    uint32_t bits, child;
    {
        const DLightNode root = load_light_node(sc.lnodes, 0);
        bits = root.bits, child = root.child1_or_light;
    }
    for (int it = 0; it < 64; ++it) {
        if (bits & 2u) return pm;
        const DLightNode n0 = load_light_node(sc.lnodes, (int)child), n1 = load_light_node(sc.lnodes, (int)child + 1);
        float c0 = node_importance(n0, p, n);
        float c1 = node_importance(n1, p, n);
        visited += 2;
        float sum = c0 + c1;
        if (sum <= 0.0f) return 0.0f;
        if ((trail & 1u) == 0u) {
            pm *= c0 / sum;
            bits = n0.bits, child = n0.child1_or_light;
        } else {
            pm *= c1 / sum;
            bits = n1.bits, child = n1.child1_or_light;
        }
        trail >>= 1;
    }
    return pm;
}


// ------------------------------------------------------------------------------------------------
// participating media (integrators/volpath/media.jl, nanovdb.jl, delta-tracking.jl:18-58, intersection.jl:422-542)
// ------------------------------------------------------------------------------------------------
HKD float hg_p(float g, float ct) {  // media.jl:36-40
    float g2 = g * g;
    float denom = 1.0f + g2 - 2.0f * g * ct;
    return (1.0f - g2) / (4.0f * PI_F * denom * sqrtf(denom));
}
HKD v3 sample_hg(float g, v3 wo, v2 u, float& pdf) {  // media.jl:51-72
    float ct;
    if (fabsf(g) < 1e-3f)
        ct = 1.0f - 2.0f * u.x;
    else {
        float g2 = g * g;
        float sq = (1.0f - g2) / (1.0f - g + 2.0f * g * u.x);
        ct = clampf((1.0f + g2 - sq * sq) / (2.0f * g), -1.0f, 1.0f);
    }
    float st = sqrtf(maxf(0.0f, 1.0f - ct * ct));
    float phi = 2.0f * PI_F * u.y;
    v3 t1, t2;
    coordinate_system(-wo, t1, t2);
    float sphi, cphi;
    jl_sincos(phi, sphi, cphi);
    v3 wi = st * cphi * t1 + st * sphi * t2 + ct * (-wo);
    wi = normalize(wi);
    pdf = hg_p(g, ct);
    return wi;
}
HKD uint64_t lcg_init(v3 o, v3 d, float t_max) {
    uint64_t ox = __float_as_uint(o.x), oy = __float_as_uint(o.y), oz = __float_as_uint(o.z), tm = __float_as_uint(t_max);
    uint64_t dx = __float_as_uint(d.x), dy = __float_as_uint(d.y), dz = __float_as_uint(d.z);
    return mix_bits(ox ^ (oy << 16) ^ (oz << 32) ^ tm) ^ mix_bits(dx ^ (dy << 16) ^ (dz << 32));
}
HKD float lcg_next(uint64_t& s) {
    s = s * 0x5DEECE66Dull + 11ull;
    float r = (float)(uint32_t)(s >> 32) * 2.3283064365386963e-10f;
    const float lim = 1.0f - 1.1920929e-7f;
    return r < lim ? r : lim;
}
HKD S4 s4max0(S4 a) { return s4(maxf(a.x, 0.0f), maxf(a.y, 0.0f), maxf(a.z, 0.0f), maxf(a.w, 0.0f)); }
// exp / log of the tracking loops through the hardware's exp2 / log2 (1 ulp each; the argument scaling adds ~|x| * 6e-8 relative):
// the free-flight and transmittance arithmetic of a path is statistical anyway (its RNG is seeded from float bit patterns,
// DESIGN.md §2), and ocml's expf / logf cost ~25 instructions per call in loops that run at a third of the lanes
#ifndef HK_MEDIA_LIBM
HKD float media_expf(float x) { return __builtin_amdgcn_exp2f(1.4426950408889634f * x); }
HKD float media_logf(float x) { return 0.6931471805599453f * __builtin_amdgcn_logf(x); }
#else
HKD float media_expf(float x) { return expf(x); }
HKD float media_logf(float x) { return logf(x); }
#endif
HKD S4 s4exp(S4 a) {
    // grey media (all four wavelengths see the same sigma): one exp, bit-identical to four
    if (a.x == a.y && a.x == a.z && a.x == a.w) {
        float e = media_expf(a.x);
        return s4(e, e, e, e);
    }
    return s4(media_expf(a.x), media_expf(a.y), media_expf(a.z), media_expf(a.w));
}
// a / y through one correctly-rounded reciprocal: each component within 1 ulp of the IEEE quotient, at a third of the
// instructions (a full-precision f32 division is ~10 VALU ops on CDNA).  Used in the tracking loops only, whose paths cannot
// be bit-reproduced across platforms anyway (their RNG streams are seeded from libm-dependent ray bits).
HKD S4 div4(S4 a, float y) {
    float r = 1.0f / y;
    return s4(a.x * r, a.y * r, a.z * r, a.w * r);
}

// RayMajorantIterator (media.jl:517-560), kept small because it lives in registers across the whole tracking loop: the
// grid pointer / resolution are re-read from the medium record, step (+1/-1) and limit (res/-1) are a sign bit per axis,
// and sigma_t is passed in by the caller (it is base_a + base_s, which the caller holds anyway).
struct MajorantIter {
    int mode;          // 0 exhausted, 1 homogeneous (not yet returned), 2 DDA; bits 8..10: axis steps negative
    float t_min, t_max;
    float next_t[3], delta_t[3];
    int voxel[3];
};
HKD void ray_bounds_intersect(v3 o, v3 d, const float* bmin, const float* bmax, float& t_enter, float& t_exit) {  // media.jl:1698-1734
    float te = -INF_F, tx = INF_F;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        float dk = comp(d, k), ok = comp(o, k);
        float inv = fabsf(dk) > 1e-10f ? 1.0f / dk : (dk >= 0 ? INF_F : -INF_F);
        float t0 = (bmin[k] - ok) * inv, t1 = (bmax[k] - ok) * inv;
        if (t0 > t1) {
            float tmp = t0;
            t0 = t1;
            t1 = tmp;
        }
        te = k == 0 ? t0 : maxf(te, t0);
        tx = k == 0 ? t1 : minf(tx, t1);
    }
    t_enter = te;
    t_exit = tx;
}
HKD MajorantIter exhausted_iter() {
    MajorantIter it;
    it.mode = 0;
    it.t_min = INF_F;
    it.t_max = -INF_F;
    return it;
}
// MM is the set of medium kinds present in the scene (bit k = HK_MEDIUM_* k): the media kernels are instantiated per set so a
// NanoVDB-only scene does not carry the registers of the RGB-grid run-time uplift (and vice versa).
#define HK_HAS_MEDIUM(MM, K) (((MM) >> (K)) & 1)
template <int MM>
HKD MajorantIter create_majorant_iterator(const DMedium& m, v3 ro, v3 rd, float t_max) {
    MajorantIter it = exhausted_iter();
    if (HK_HAS_MEDIUM(MM, HK_MEDIUM_HOMOGENEOUS) && m.kind == HK_MEDIUM_HOMOGENEOUS) {
        it.mode = (0.0f >= t_max) ? 0 : 1;
        it.t_min = 0.0f;
        it.t_max = t_max;
        return it;
    }
    v3 o = ro, d = rd;
    if ((HK_HAS_MEDIUM(MM, HK_MEDIUM_GRID) && m.kind == HK_MEDIUM_GRID) || (HK_HAS_MEDIUM(MM, HK_MEDIUM_RGB_GRID) && m.kind == HK_MEDIUM_RGB_GRID)) {
        const float* M = m.r2m;
        o = mk3(M[0] * ro.x + M[1] * ro.y + M[2] * ro.z + M[3], M[4] * ro.x + M[5] * ro.y + M[6] * ro.z + M[7], M[8] * ro.x + M[9] * ro.y + M[10] * ro.z + M[11]);
        d = mk3(M[0] * rd.x + M[1] * rd.y + M[2] * rd.z, M[4] * rd.x + M[5] * rd.y + M[6] * rd.z, M[8] * rd.x + M[9] * rd.y + M[10] * rd.z);
        if (d.x * d.x + d.y * d.y + d.z * d.z < 1e-20f) return it;
    }
    float t_enter, t_exit;
    ray_bounds_intersect(o, d, m.bmin, m.bmax, t_enter, t_exit);
    t_enter = maxf(t_enter, 0.0f);
    t_exit = minf(t_exit, t_max);
    if (t_enter >= t_exit) return it;
    // create_dda_iterator (media.jl:229-340)
    int mode = 2;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        int res = m.mres[k];
        float diag = m.bmax[k] - m.bmin[k];
        float go = (comp(o, k) - m.bmin[k]) / diag;
        float inv_diag = fabsf(diag) > 1e-10f ? 1.0f / diag : 0.0f;
        float gd = comp(d, k) * inv_diag;
        float gi = go + gd * t_enter;
        int v = clampi((int)floorf(gi * (float)res), 0, res - 1);
        it.voxel[k] = v;
        it.delta_t[k] = fabsf(gd) > 1e-10f ? 1.0f / (fabsf(gd) * (float)res) : INF_F;
        if (gd >= 0.0f) {
            float nvp = (float)(v + 1) / (float)res;
            it.next_t[k] = gd > 1e-10f ? t_enter + (nvp - gi) / gd : INF_F;
        } else {
            float nvp = (float)v / (float)res;
            it.next_t[k] = gd < -1e-10f ? t_enter + (nvp - gi) / gd : INF_F;
            mode |= 0x100 << k;
        }
    }
    it.t_min = t_enter;
    it.t_max = t_exit;
    it.mode = mode;
    return it;
}
// sigma_t: the medium's sigma_a + sigma_s at the path's wavelengths (base_a + base_s); RGBGrid media use a unit sigma_t, the
// scale is in their majorant grid (media.jl:1408-1420)
template <int MM>
HKD bool majorant_next(MajorantIter& it, const DMedium& m, S4 sigma_t, float& seg_t_min, float& seg_t_max, S4& sigma_maj) {  // media.jl:625-729
    const int mode = it.mode & 0xff;
    if (mode == 0) return false;
    if (HK_HAS_MEDIUM(MM, HK_MEDIUM_HOMOGENEOUS) && mode == 1) {
        it.mode = 0;
        if (it.t_min >= it.t_max) return false;
        seg_t_min = it.t_min;
        seg_t_max = it.t_max;
        sigma_maj = sigma_t;
        return true;
    }
    if (it.t_min >= it.t_max) {
        it.mode = 0;
        return false;
    }
    if (HK_HAS_MEDIUM(MM, HK_MEDIUM_RGB_GRID) && m.kind == HK_MEDIUM_RGB_GRID) sigma_t = s4(1.0f);
    // one DDA step without a branch (a divergent wave would walk all three axis arms, each behind its own exec-mask bookkeeping):
    // the stepping axis is the one media.jl:676-690 picks, the crossing time of that axis advances by ONE addition as there
    const float tx = it.next_t[0], ty = it.next_t[1], tz = it.next_t[2];
    const bool lxy = tx < ty, lxz = tx < tz, lyz = ty < tz;
    const bool a0 = lxy & lxz, a1 = (!lxy) & lyz;   // axis 0, axis 1, else axis 2
    const float nt = a0 ? tx : (a1 ? ty : tz);
    const float stm = minf(nt, it.t_max);
    const int rx = m.mres[0], ry = m.mres[1], rz = m.mres[2];
    const int vx = it.voxel[0], vy = it.voxel[1], vz = it.voxel[2];
    const float rho = m.majorant[vx + rx * (vy + ry * vz)];
    seg_t_min = it.t_min;
    seg_t_max = stm;
    sigma_maj = sigma_t * rho;
    const bool neg = (it.mode & (a0 ? 0x100 : (a1 ? 0x200 : 0x400))) != 0;
    const int v = (a0 ? vx : (a1 ? vy : vz)) + (neg ? -1 : 1);
    const int lim = neg ? -1 : (a0 ? rx : (a1 ? ry : rz));
    const float d0 = it.delta_t[0], d1 = it.delta_t[1], d2 = it.delta_t[2];   // (named: an indexed pick makes the compiler keep the array in LDS)
    const float s = nt + (a0 ? d0 : (a1 ? d1 : d2));
    it.voxel[0] = a0 ? v : vx;
    it.voxel[1] = a1 ? v : vy;
    it.voxel[2] = (a0 | a1) ? vz : v;
    it.next_t[0] = a0 ? s : tx;
    it.next_t[1] = a1 ? s : ty;
    it.next_t[2] = (a0 | a1) ? tz : s;
    const bool out = v == lim;
    it.mode = out ? 0 : it.mode;
    it.t_min = out ? it.t_max : stm;
    return true;
}

// Fast-forward over majorant cells whose value is exactly 0 (65 % of the cells of a cloud): what majorant_next + the caller's
// "sigma_maj < 1e-10 -> next segment" do for such a cell is ONLY the DDA step and ++segi, so the same float operations are done
// here in a tight per-lane loop that reads one bit per cell (a 4 KB table for 32^3 cells, L1-resident) instead of one float from
// the majorant grid per outer iteration of the tracking state machine.  State afterwards == state after the same number of
// majorant_next calls (bit for bit: hk_test_medium mode 2).
template <int MM>
HKD void majorant_skip_zero(MajorantIter& it, const DMedium& m, int& segi) {
    if ((it.mode & 0xff) != 2 || m.maj_zero == nullptr) return;
    const int rx = m.mres[0], ry = m.mres[1], rz = m.mres[2];
    while (segi < 256 && it.t_min < it.t_max) {
        const int cell = it.voxel[0] + rx * (it.voxel[1] + ry * it.voxel[2]);
        if (((m.maj_zero[cell >> 5] >> (cell & 31)) & 1u) == 0u) break;
        const float tx = it.next_t[0], ty = it.next_t[1], tz = it.next_t[2];
        const int axis = (tx < ty) ? ((tx < tz) ? 0 : 2) : ((ty < tz) ? 1 : 2);
        const float nt = axis == 0 ? tx : (axis == 1 ? ty : tz);
        it.t_min = minf(nt, it.t_max);
        bool out;
        if (axis == 0) {
            const bool neg = it.mode & 0x100;
            it.voxel[0] += neg ? -1 : 1;
            it.next_t[0] += it.delta_t[0];
            out = it.voxel[0] == (neg ? -1 : rx);
        } else if (axis == 1) {
            const bool neg = it.mode & 0x200;
            it.voxel[1] += neg ? -1 : 1;
            it.next_t[1] += it.delta_t[1];
            out = it.voxel[1] == (neg ? -1 : ry);
        } else {
            const bool neg = it.mode & 0x400;
            it.voxel[2] += neg ? -1 : 1;
            it.next_t[2] += it.delta_t[2];
            out = it.voxel[2] == (neg ? -1 : rz);
        }
        ++segi;
        if (out) {
            it.mode = 0;
            it.t_min = it.t_max;
            return;
        }
    }
}

// NanoVDB lookups.  The tree walk (nanovdb.jl:315-388, hk_nanovdb.h) runs on the HOST, once, at scene upload: it flattens the
// tree over the grid's index bounding box (+1 block of margin) into a block table, one 8-byte entry per 8^3 block = {leaf
// offset, or 0 and the block's constant tile / background value}.  On the device a voxel fetch is table entry -> leaf value:
// two dependent loads instead of six, and no tree-walk code (and registers) in the tracking kernels.  Blocks outside the
// table hold the background value (the upload checks that the margin does).  Values are those the walk returns.
typedef float hk_f2u __attribute__((ext_vector_type(2), aligned(4)));
struct NvBlock {
    uint32_t leaf_off;   // 1-based byte offset of the leaf node, 0 => constant `value` for the whole block
    float value;
};
HKD NvBlock nv_find_block(const DMedium& m, int kx, int ky, int kz) {  // block coordinates (x>>3, y>>3, z>>3)
    NvBlock r;
    const int bx = kx - m.nvb_min[0], by = ky - m.nvb_min[1], bz = kz - m.nvb_min[2];
    if ((unsigned)bx < (unsigned)m.nvb_dim[0] && (unsigned)by < (unsigned)m.nvb_dim[1] && (unsigned)bz < (unsigned)m.nvb_dim[2]) {
        uint2 e = m.nv_blocks[(size_t)bz + (size_t)m.nvb_dim[2] * ((size_t)by + (size_t)m.nvb_dim[1] * (size_t)bx)];
        r.leaf_off = e.x;
        r.value = __uint_as_float(e.y);
    } else {
        r.leaf_off = 0u;
        r.value = m.nv_background;
    }
    return r;
}
HKD float nv_leaf_value(const DMedium& m, uint32_t leaf_off, int n_leaf) { return hknv::f32(m.nvdb, (long long)leaf_off + 96 + (long long)n_leaf * 4); }
// BRICKS_ONLY: the caller knows that the medium has dense bricks (DScene::grey_bricks), the tree-walk paths are not instantiated
template <bool BRICKS_ONLY = false>
HKD float sample_nanovdb_density(const DMedium& m, v3 p) {  // nanovdb.jl:400-469
    float px = p.x - m.vec[0], py = p.y - m.vec[1], pz = p.z - m.vec[2];
    float fxi = m.inv_mat[0] * px + m.inv_mat[1] * py + m.inv_mat[2] * pz;
    float fyi = m.inv_mat[3] * px + m.inv_mat[4] * py + m.inv_mat[5] * pz;
    float fzi = m.inv_mat[6] * px + m.inv_mat[7] * py + m.inv_mat[8] * pz;
    float flx = floorf(fxi), fly = floorf(fyi), flz = floorf(fzi);
    int ix = (int)flx, iy = (int)fly, iz = (int)flz;
    float fx = fxi - (float)ix, fy = fyi - (float)iy, fz = fzi - (float)iz;
    float v000, v001, v010, v011, v100, v101, v110, v111;
    if (BRICKS_ONLY || m.nv_bricks) {
        // dense bricks WITH A HALO (the upload materialises them when the index bounding box is small enough): brick = the block's 8^3
        // voxels plus the first plane of its +x / +y / +z neighbours, 9^3 floats, z fastest — the eight taps of ANY voxel are in the
        // brick of its block: four 8-byte loads (4-byte aligned) at one computed address, no dependent table lookup, no second path
        // for the 33 % of the voxels whose taps straddle two blocks.  Same voxel values as the tree walk.
        const int bx = (ix >> 3) - m.nvb_min[0], by = (iy >> 3) - m.nvb_min[1], bz = (iz >> 3) - m.nvb_min[2];
        if ((unsigned)bx < (unsigned)m.nvb_dim[0] && (unsigned)by < (unsigned)m.nvb_dim[1] && (unsigned)bz < (unsigned)m.nvb_dim[2]) {
            const float* b = m.nv_bricks + ((size_t)(bz + m.nvb_dim[2] * (by + m.nvb_dim[1] * bx)) * 729u + (size_t)((ix & 7) * 81 + (iy & 7) * 9 + (iz & 7)));
            const hk_f2u p00 = *(const hk_f2u*)(b), p01 = *(const hk_f2u*)(b + 9), p10 = *(const hk_f2u*)(b + 81), p11 = *(const hk_f2u*)(b + 90);
            v000 = p00.x, v001 = p00.y, v010 = p01.x, v011 = p01.y, v100 = p10.x, v101 = p10.y, v110 = p11.x, v111 = p11.y;
        } else   // the table's outermost blocks hold the background (checked at upload), so a base voxel outside it has background taps only
            v000 = v001 = v010 = v011 = v100 = v101 = v110 = v111 = m.nv_background;
    } else if (!BRICKS_ONLY && (ix & 7) != 7 && (iy & 7) != 7 && (iz & 7) != 7) {
        // all eight taps in one 8^3 block (2 of 3 lookups): one table entry, eight independent leaf loads
        NvBlock c = nv_find_block(m, ix >> 3, iy >> 3, iz >> 3);
        if (c.leaf_off == 0u)
            v000 = v001 = v010 = v011 = v100 = v101 = v110 = v111 = c.value;
        else {
            int n = ((ix & 7) << 6) | ((iy & 7) << 3) | (iz & 7);
            v000 = nv_leaf_value(m, c.leaf_off, n);
            v001 = nv_leaf_value(m, c.leaf_off, n + 1);
            v010 = nv_leaf_value(m, c.leaf_off, n + 8);
            v011 = nv_leaf_value(m, c.leaf_off, n + 9);
            v100 = nv_leaf_value(m, c.leaf_off, n + 64);
            v101 = nv_leaf_value(m, c.leaf_off, n + 65);
            v110 = nv_leaf_value(m, c.leaf_off, n + 72);
            v111 = nv_leaf_value(m, c.leaf_off, n + 73);
        }
    } else if (BRICKS_ONLY) {
        v000 = v001 = v010 = v011 = v100 = v101 = v110 = v111 = 0.0f;   // (not reached)
    } else {
        v000 = v001 = v010 = v011 = v100 = v101 = v110 = v111 = 0.0f;
#pragma unroll 1
        for (int tap = 0; tap < 8; ++tap) {
            int x = ix + (tap >> 2), y = iy + ((tap >> 1) & 1), z = iz + (tap & 1);
            NvBlock c = nv_find_block(m, x >> 3, y >> 3, z >> 3);
            float v = c.leaf_off == 0u ? c.value : nv_leaf_value(m, c.leaf_off, ((x & 7) << 6) | ((y & 7) << 3) | (z & 7));
            v000 = tap == 0 ? v : v000;
            v001 = tap == 1 ? v : v001;
            v010 = tap == 2 ? v : v010;
            v011 = tap == 3 ? v : v011;
            v100 = tap == 4 ? v : v100;
            v101 = tap == 5 ? v : v101;
            v110 = tap == 6 ? v : v110;
            v111 = tap == 7 ? v : v111;
        }
    }
    float fx1 = 1.0f - fx, fy1 = 1.0f - fy, fz1 = 1.0f - fz;
    float v00 = v000 * fz1 + v001 * fz, v01 = v010 * fz1 + v011 * fz;
    float v10 = v100 * fz1 + v101 * fz, v11 = v110 * fz1 + v111 * fz;
    float v0 = v00 * fy1 + v01 * fy, v1 = v10 * fy1 + v11 * fy;
    return v0 * fx1 + v1 * fx;
}
HKD float sample_grid_density(const DMedium& m, v3 pm) {  // media.jl:1527-1575
    float pn0 = (pm.x - m.bmin[0]) / (m.bmax[0] - m.bmin[0]), pn1 = (pm.y - m.bmin[1]) / (m.bmax[1] - m.bmin[1]), pn2 = (pm.z - m.bmin[2]) / (m.bmax[2] - m.bmin[2]);
    if (pn0 < 0.0f || pn1 < 0.0f || pn2 < 0.0f || pn0 > 1.0f || pn1 > 1.0f || pn2 > 1.0f) return 0.0f;
    int nx = m.res[0], ny = m.res[1], nz = m.res[2];
    float gx = pn0 * (float)nx + 0.5f, gy = pn1 * (float)ny + 0.5f, gz = pn2 * (float)nz + 0.5f;
    int ix = clampi((int)floorf(gx), 1, nx - 1), iy = clampi((int)floorf(gy), 1, ny - 1), iz = clampi((int)floorf(gz), 1, nz - 1);
    float fx = clampf(gx - (float)ix, 0.0f, 1.0f), fy = clampf(gy - (float)iy, 0.0f, 1.0f), fz = clampf(gz - (float)iz, 0.0f, 1.0f);
    const float* D = m.density;
    size_t sx = 1, sy = (size_t)nx, sz = (size_t)nx * ny;
    size_t base = (size_t)(ix - 1) + sy * (size_t)(iy - 1) + sz * (size_t)(iz - 1);
    float d000 = D[base], d100 = D[base + sx], d010 = D[base + sy], d110 = D[base + sx + sy];
    float d001 = D[base + sz], d101 = D[base + sx + sz], d011 = D[base + sy + sz], d111 = D[base + sx + sy + sz];
    float fx1 = 1.0f - fx;
    float d00 = d000 * fx1 + d100 * fx, d10 = d010 * fx1 + d110 * fx, d01 = d001 * fx1 + d101 * fx, d11 = d011 * fx1 + d111 * fx;
    float fy1 = 1.0f - fy;
    float d0 = d00 * fy1 + d10 * fy, d1 = d01 * fy1 + d11 * fy;
    return d0 * (1.0f - fz) + d1 * fz;
}
struct MediumProps {
    S4 sigma_a, sigma_s, Le;
    float g;
};
// sigma_a/sigma_s spectra are constant per medium and wavelength set: callers evaluate them once per ray
// (`base_*`) and sample_point only fetches the density (same arithmetic: uplift * d)
HKD float4 rgb_grid_at(const float4* g, const DMedium& m, int ix, int iy, int iz) {
    return g[(size_t)(ix - 1) + (size_t)m.res[0] * ((size_t)(iy - 1) + (size_t)m.res[1] * (size_t)(iz - 1))];
}
HKD float4 lerp4(float4 a, float wa, float4 b, float wb) { return make_float4(a.x * wa + b.x * wb, a.y * wa + b.y * wb, a.z * wa + b.z * wb, 0.0f); }
HKD float4 sample_rgb_grid(const float4* g, const DMedium& m, v3 pn) {  // media.jl:1282-1324
    if (pn.x < 0.0f || pn.y < 0.0f || pn.z < 0.0f || pn.x > 1.0f || pn.y > 1.0f || pn.z > 1.0f) return make_float4(0, 0, 0, 0);
    int nx = m.res[0], ny = m.res[1], nz = m.res[2];
    float gx = pn.x * (float)nx + 0.5f, gy = pn.y * (float)ny + 0.5f, gz = pn.z * (float)nz + 0.5f;
    int ix = clampi((int)floorf(gx), 1, nx - 1), iy = clampi((int)floorf(gy), 1, ny - 1), iz = clampi((int)floorf(gz), 1, nz - 1);
    float fx = clampf(gx - (float)ix, 0.0f, 1.0f), fy = clampf(gy - (float)iy, 0.0f, 1.0f), fz = clampf(gz - (float)iz, 0.0f, 1.0f);
    float fx1 = 1.0f - fx, fy1 = 1.0f - fy;
    float4 c00 = lerp4(rgb_grid_at(g, m, ix, iy, iz), fx1, rgb_grid_at(g, m, ix + 1, iy, iz), fx);
    float4 c10 = lerp4(rgb_grid_at(g, m, ix, iy + 1, iz), fx1, rgb_grid_at(g, m, ix + 1, iy + 1, iz), fx);
    float4 c01 = lerp4(rgb_grid_at(g, m, ix, iy, iz + 1), fx1, rgb_grid_at(g, m, ix + 1, iy, iz + 1), fx);
    float4 c11 = lerp4(rgb_grid_at(g, m, ix, iy + 1, iz + 1), fx1, rgb_grid_at(g, m, ix + 1, iy + 1, iz + 1), fx);
    return lerp4(lerp4(c00, fy1, c10, fy), 1.0f - fz, lerp4(c01, fy1, c11, fy), fz);
}
template <int MM>
HKD MediumProps sample_point(const DTables& T, S4 lambda, const DMedium& m, S4 base_a, S4 base_s, S4 base_Le, v3 p) {
    MediumProps mp;
    mp.g = m.g;
    if (HK_HAS_MEDIUM(MM, HK_MEDIUM_HOMOGENEOUS) && m.kind == HK_MEDIUM_HOMOGENEOUS) {
        mp.sigma_a = base_a;
        mp.sigma_s = base_s;
        mp.Le = base_Le;
        return mp;
    }
    if (HK_HAS_MEDIUM(MM, HK_MEDIUM_RGB_GRID) && m.kind == HK_MEDIUM_RGB_GRID) {  // media.jl:1327-1370: per-point uplift_rgb_unbounded of the interpolated RGB
        const float* M = m.r2m;
        v3 pm = mk3(M[0] * p.x + M[1] * p.y + M[2] * p.z + M[3], M[4] * p.x + M[5] * p.y + M[6] * p.z + M[7], M[8] * p.x + M[9] * p.y + M[10] * p.z + M[11]);
        v3 pn = mk3((pm.x - m.bmin[0]) / (m.bmax[0] - m.bmin[0]), (pm.y - m.bmin[1]) / (m.bmax[1] - m.bmin[1]), (pm.z - m.bmin[2]) / (m.bmax[2] - m.bmin[2]));
        float4 a = m.rgb_a ? sample_rgb_grid(m.rgb_a, m, pn) : make_float4(1, 1, 1, 1);
        float4 sc = m.rgb_s ? sample_rgb_grid(m.rgb_s, m, pn) : make_float4(1, 1, 1, 1);
        mp.sigma_a = eval_scaled(coef_unbounded(T, a.x, a.y, a.z), lambda) * m.sigma_scale;
        mp.sigma_s = eval_scaled(coef_unbounded(T, sc.x, sc.y, sc.z), lambda) * m.sigma_scale;
        mp.Le = s4(0.0f);
        if (m.rgb_Le && m.Le_scale > 0.0f) {
            float4 e = sample_rgb_grid(m.rgb_Le, m, pn);
            mp.Le = eval_scaled(coef_unbounded(T, e.x, e.y, e.z), lambda) * m.Le_scale;
        }
        return mp;
    }
    float d;
    if (HK_HAS_MEDIUM(MM, HK_MEDIUM_GRID) && (m.kind == HK_MEDIUM_GRID || !HK_HAS_MEDIUM(MM, HK_MEDIUM_NANOVDB))) {
        const float* M = m.r2m;
        d = sample_grid_density(m, mk3(M[0] * p.x + M[1] * p.y + M[2] * p.z + M[3], M[4] * p.x + M[5] * p.y + M[6] * p.z + M[7], M[8] * p.x + M[9] * p.y + M[10] * p.z + M[11]));
    } else
        d = sample_nanovdb_density(m, p);
    mp.sigma_a = base_a * d;
    mp.sigma_s = base_s * d;
    mp.Le = s4(0.0f);
    return mp;
}
// density at p of a density-scaled medium (Grid / NanoVDB): what sample_point multiplies the medium's spectra with
template <int MM, bool BRICKS_ONLY = false>
HKD float sample_density(const DMedium& m, v3 p) {
    if (HK_HAS_MEDIUM(MM, HK_MEDIUM_GRID) && (m.kind == HK_MEDIUM_GRID || !HK_HAS_MEDIUM(MM, HK_MEDIUM_NANOVDB))) {
        const float* M = m.r2m;
        return sample_grid_density(m, mk3(M[0] * p.x + M[1] * p.y + M[2] * p.z + M[3], M[4] * p.x + M[5] * p.y + M[6] * p.z + M[7], M[8] * p.x + M[9] * p.y + M[10] * p.z + M[11]));
    }
    return sample_nanovdb_density<BRICKS_ONLY>(m, p);
}
// GREY media: sigma_a and sigma_s are flat spectra (sigmoid coefficients c0 = c1 = 0: what a grey RGB uplifts to, and what the
// BOMEX example's RGBSpectrum(0) / RGBSpectrum(1) are) — the value is the same at every wavelength, bit for bit
HKD float eval_flat(float4 c) { return c.w == 0.0f ? 0.0f : c.w * poly_eval(0.0f, 0.0f, c.z, 500.0f); }
// average() of an S4 whose four components are all v, in average()'s own operation order
HKD float average_flat(float v) { return (((v + v) + v) + v) / 4.0f; }
}  // namespace hkd

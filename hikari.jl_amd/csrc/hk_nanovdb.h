// hk_nanovdb.h — NanoVDB tree walk (nanovdb.jl:315-388) shared by the device code and the host-side block-table builder.
// Offsets are 1-based byte positions, like the reference's fields.
#pragma once
#include <hip/hip_runtime.h>

#define HK_HD __host__ __device__ __forceinline__

namespace hknv {

HK_HD float f32(const unsigned char* b, long long off1) { return *reinterpret_cast<const float*>(b + (((off1 - 1) >> 2) << 2)); }
HK_HD long long i64(const unsigned char* b, long long off1) { return *reinterpret_cast<const long long*>(b + (((off1 - 1) >> 3) << 3)); }
HK_HD bool mask(const unsigned char* b, long long mask_off1, int n) { return ((b[mask_off1 - 1 + (n >> 3)] >> (n & 7)) & 1) != 0; }

// Walk root -> upper 32^3 -> lower 16^3 down to the 8^3 block that holds (x,y,z).  Returns the 1-based leaf offset, or 0 with
// `value` = the constant (tile / background) value of the whole block.
HK_HD long long find_block(const unsigned char* b, long long root, int root_table_size, int x, int y, int z, float& value) {
    value = 0.0f;
    unsigned xu = (unsigned)x, yu = (unsigned)y, zu = (unsigned)z;
    unsigned long long key = (unsigned long long)((zu >> 12) & 0x1fffff) | ((unsigned long long)((yu >> 12) & 0x1fffff) << 21) | ((unsigned long long)((xu >> 12) & 0x1fffff) << 42);
    long long tile = 0;
    bool found = false;
    for (int i = 0; i < root_table_size; ++i) {
        long long t_off = root + 64 + (long long)i * 32;
        if ((unsigned long long)i64(b, t_off) == key) {
            found = true;
            tile = t_off;
            break;
        }
    }
    if (!found) {
        value = f32(b, root + 28);
        return 0;
    }
    long long child = i64(b, tile + 8);
    if (child == 0) {
        value = f32(b, tile + 20);
        return 0;
    }
    long long upper = root + child;
    int n_upper = (int)(((xu >> 7) & 31) << 10) | (int)(((yu >> 7) & 31) << 5) | (int)((zu >> 7) & 31);
    if (!mask(b, upper + 4128, n_upper)) {
        value = f32(b, upper + 8256 + (long long)n_upper * 8);
        return 0;
    }
    long long lower = upper + i64(b, upper + 8256 + (long long)n_upper * 8);
    int n_lower = (int)(((xu >> 3) & 15) << 8) | (int)(((yu >> 3) & 15) << 4) | (int)((zu >> 3) & 15);
    if (!mask(b, lower + 544, n_lower)) {
        value = f32(b, lower + 1088 + (long long)n_lower * 8);
        return 0;
    }
    return lower + i64(b, lower + 1088 + (long long)n_lower * 8);
}

}  // namespace hknv

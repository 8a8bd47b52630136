// bvh_build.h — host-side acceleration-structure builders of the HIP library.
#pragma once
#include <cstdint>
#include <vector>

#include "hikari_mi355x.h"

namespace hk {

struct BVHNode {
    float lo0[3], hi0[3], lo1[3], hi1[3];
    int c0, c1;
};
struct BVH {
    std::vector<BVHNode> nodes;
    std::vector<int> leaf_prims;  // original triangle index per leaf slot
    int root_ref = 0;
    int max_depth = 0;
    float lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
};
void build_bvh(const float* positions, int n_tris, BVH& out, int leaf_size = 4, int bins = 32);   // leaf_size: triangles per leaf, 1 .. 8; bins: SAH bins per axis, 4 .. 64
// (32 bins since round 5: the 10^6-triangle scene's traversal - 1.7 %, its any-hit casts - 1.5 % against 16, 64 no better, 8 + 1 %; the small scenes unchanged)

// Light BVH (lights/bvh-light-sampler.jl:283-466).  Node = 16 floats/uints as uploaded to the device.
struct LightBVHNodeH {
    float bmin[3], bmax[3], w[3];
    float phi, cos_o, cos_e;
    uint32_t bits;             // bit0 two_sided, bit1 leaf
    uint32_t child1_or_light;  // 1-based
    uint32_t pad[2];
};
struct LightBVH {
    std::vector<LightBVHNodeH> nodes;
    std::vector<uint32_t> bit_trails;   // per light, 0xFFFFFFFF = not in the BVH
    std::vector<int32_t> infinite;      // 1-based flat indices
    int num_bvh = 0;
};
// max_poly[i]: max_value of light i's sigmoid polynomial (needed for RGBIlluminantSpectrum luminance)
void build_light_bvh(const hk_light* lights, int n, LightBVH& out);

// sigmoid-polynomial helpers shared by the host-side bakers (spectral/rgb2spec.jl:17-53, 85-167)
struct RGB2Spec {
    int res = 0;
    const float* scale = nullptr;
    const float* coeffs = nullptr;
};
void rgb_to_coeffs(const RGB2Spec& t, float r, float g, float b, float out[3]);
float poly_eval(const float c[3], float lambda);
float poly_max(const float c[3]);

}  // namespace hk

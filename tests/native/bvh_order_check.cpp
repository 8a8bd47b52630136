// Host-only check of hikari.jl_amd/csrc/bvh_build.cpp (built by tests/test_abi_and_host.py::test_bvh_builder_invariants with g++):
// breadth-first numbering (root 0, every child index above its parent, levels never decrease along the array — what the LDS node cache
// of k_trace_lean relies on), every node referenced once, every triangle in exactly one leaf of <= 4, child boxes enclose their
// triangles, depth within the 32-entry stack.   usage: bvh_order_check <n_tris> <seed>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "bvh_build.h"

static unsigned rng_state;
static float frand() {
    rng_state = rng_state * 1664525u + 1013904223u;
    return (float)(rng_state >> 8) * (1.0f / 16777216.0f);
}
#define CHECK(c)                                                  \
    if (!(c)) {                                                   \
        std::printf("FAILED %s (line %d)\n", #c, __LINE__);       \
        return 1;                                                 \
    }

int main(int argc, char** argv) {
    const int n = argc > 1 ? std::atoi(argv[1]) : 1000;
    rng_state = argc > 2 ? (unsigned)std::atoi(argv[2]) : 1u;
    std::vector<float> pos(9 * (size_t)n);
    for (int t = 0; t < n; ++t) {
        // clustered: most triangles small, a few huge, some exact duplicates (the builder's median fallback)
        float c[3] = {frand() * 10.0f, frand() * 10.0f, frand() * (t % 7 == 0 ? 0.0f : 10.0f)};
        float size = t % 97 == 0 ? 8.0f : 0.05f;
        for (int v = 0; v < 3; ++v)
            for (int k = 0; k < 3; ++k) pos[9 * (size_t)t + 3 * v + k] = (t % 13 == 5 && t > 0) ? pos[9 * (size_t)(t - 1) + 3 * v + k] : c[k] + size * (frand() - 0.5f);
    }
    hk::BVH bvh;
    hk::build_bvh(pos.data(), n, bvh);
    const int nn = (int)bvh.nodes.size();
    if (n <= 4) {
        CHECK(nn == 0 && bvh.root_ref < 0);
        std::printf("ok leaf-only\n");
        return 0;
    }
    CHECK(bvh.root_ref == 0 && nn >= 1);
    CHECK(bvh.max_depth <= 30);
    std::vector<int> level(nn, -1), refs(nn, 0);
    std::vector<int> seen(bvh.leaf_prims.size(), 0);
    level[0] = 0;
    int deepest = 0;
    for (int i = 0; i < nn; ++i) {
        CHECK(level[i] >= 0);                                    // a parent precedes its children
        if (i > 0) CHECK(level[i] >= level[i - 1]);              // breadth-first: levels never decrease along the array
        const hk::BVHNode& nd = bvh.nodes[i];
        const int child[2] = {nd.c0, nd.c1};
        const float* lo[2] = {nd.lo0, nd.lo1};
        const float* hi[2] = {nd.hi0, nd.hi1};
        for (int s = 0; s < 2; ++s) {
            if (child[s] >= 0) {
                CHECK(child[s] > i && child[s] < nn);
                ++refs[child[s]];
                level[child[s]] = level[i] + 1;
                if (level[i] + 1 > deepest) deepest = level[i] + 1;
            } else {
                const int ref = ~child[s], first = ref >> 3, count = (ref & 7) + 1;
                CHECK(count <= 4 && first >= 0 && first + count <= (int)bvh.leaf_prims.size());
                for (int k = first; k < first + count; ++k) {
                    ++seen[k];
                    const int t = bvh.leaf_prims[k];
                    CHECK(t >= 0 && t < n);
                    for (int v = 0; v < 3; ++v)
                        for (int a = 0; a < 3; ++a) CHECK(pos[9 * (size_t)t + 3 * v + a] >= lo[s][a] && pos[9 * (size_t)t + 3 * v + a] <= hi[s][a]);
                }
            }
        }
    }
    for (int i = 1; i < nn; ++i) CHECK(refs[i] == 1);
    CHECK((int)bvh.leaf_prims.size() == n);
    std::vector<int> tri_seen(n, 0);
    for (size_t k = 0; k < seen.size(); ++k) {
        CHECK(seen[k] == 1);
        ++tri_seen[bvh.leaf_prims[k]];
    }
    for (int t = 0; t < n; ++t) CHECK(tri_seen[t] == 1);
    CHECK(deepest + 1 <= 32 && bvh.max_depth >= deepest);
    std::printf("ok nodes %d depth %d\n", nn, bvh.max_depth);
    return 0;
}

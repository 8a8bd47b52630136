// bvh_build.cpp — host-side BVH2 builder for the traversal kernels.
//
// Replaces what the reference gets from Raycore's TLAS/BVH (src/scene.jl:120-149 `sync!`, call sites
// src/integrators/volpath/intersection.jl:200,225,323,703).  Binned SAH (32 bins; 16 until round 5), leaves of <= 4
// triangles, depth bounded so a per-lane stack of HK_LDS_STACK entries can never overflow: a split
// is only taken if both children can still be finished by median splits inside the depth budget.
// Output layout: DNode (64 B, both child boxes inline) + leaf-ordered 48-B triangles.
#include "bvh_build.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <limits>

namespace hk {

namespace {
struct Box {
    float lo[3], hi[3];
    void reset() {
        for (int k = 0; k < 3; ++k) {
            lo[k] = std::numeric_limits<float>::infinity();
            hi[k] = -std::numeric_limits<float>::infinity();
        }
    }
    void grow(const float* p) {
        for (int k = 0; k < 3; ++k) {
            lo[k] = std::min(lo[k], p[k]);
            hi[k] = std::max(hi[k], p[k]);
        }
    }
    void grow(const Box& b) {
        for (int k = 0; k < 3; ++k) {
            lo[k] = std::min(lo[k], b.lo[k]);
            hi[k] = std::max(hi[k], b.hi[k]);
        }
    }
    float half_area() const {
        float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        if (!(dx >= 0)) return 0.0f;
        return dx * dy + dy * dz + dz * dx;
    }
};

int ceil_log2(int v) {
    int l = 0;
    while ((1 << l) < v) ++l;
    return l;
}

struct Builder {
    const float* pos;
    std::vector<Box> tb;
    std::vector<float> cent;  // 3 per tri
    std::vector<int> idx;
    BVH& out;
    int max_depth_seen = 0;
    int LEAF = 4;
    static constexpr int MAXBINS = 64;
    int NBINS = 16;
    static constexpr int DEPTH_BUDGET = 30;  // < HK_LDS_STACK

    Builder(const float* p, int n, BVH& o) : pos(p), out(o) {
        tb.resize(n);
        cent.resize(3 * (size_t)n);
        idx.resize(n);
        for (int i = 0; i < n; ++i) {
            tb[i].reset();
            for (int v = 0; v < 3; ++v) tb[i].grow(p + 9 * (size_t)i + 3 * v);
            for (int k = 0; k < 3; ++k) cent[3 * (size_t)i + k] = 0.5f * (tb[i].lo[k] + tb[i].hi[k]);
            idx[i] = i;
        }
    }
    // levels a median-split subtree with `count` triangles still needs (leaf capacity LEAF)
    int levels_needed(int count) const { return count <= LEAF ? 0 : ceil_log2((count + LEAF - 1) / LEAF); }

    int make_leaf(int first, int count) {
        int start = (int)out.leaf_prims.size();
        for (int i = 0; i < count; ++i) out.leaf_prims.push_back(idx[first + i]);
        return ~((start << 3) | (count - 1));
    }
    // returns child ref; box = bounds of the range
    int build(int first, int count, int depth, Box& box) {
        box.reset();
        Box cb;
        cb.reset();
        for (int i = first; i < first + count; ++i) {
            box.grow(tb[idx[i]]);
            cb.grow(&cent[3 * (size_t)idx[i]]);
        }
        max_depth_seen = std::max(max_depth_seen, depth);
        if (count <= LEAF) return make_leaf(first, count);
        // binned SAH
        int best_axis = -1, best_split = -1;
        float best_cost = std::numeric_limits<float>::infinity();
        for (int axis = 0; axis < 3; ++axis) {
            float ext = cb.hi[axis] - cb.lo[axis];
            if (!(ext > 0)) continue;
            Box bins[MAXBINS];
            int cnt[MAXBINS] = {0};
            for (auto& b : bins) b.reset();
            float scale = NBINS / ext;
            for (int i = first; i < first + count; ++i) {
                int b = (int)((cent[3 * (size_t)idx[i] + axis] - cb.lo[axis]) * scale);
                b = std::min(std::max(b, 0), NBINS - 1);
                bins[b].grow(tb[idx[i]]);
                cnt[b]++;
            }
            float right_area[MAXBINS];
            int right_cnt[MAXBINS];
            Box acc;
            acc.reset();
            int c = 0;
            for (int b = NBINS - 1; b >= 1; --b) {
                acc.grow(bins[b]);
                c += cnt[b];
                right_area[b] = acc.half_area();
                right_cnt[b] = c;
            }
            acc.reset();
            c = 0;
            for (int b = 0; b < NBINS - 1; ++b) {
                acc.grow(bins[b]);
                c += cnt[b];
                if (c == 0 || right_cnt[b + 1] == 0) continue;
                float cost = acc.half_area() * (float)c + right_area[b + 1] * (float)right_cnt[b + 1];
                if (cost < best_cost) {
                    best_cost = cost;
                    best_axis = axis;
                    best_split = b;
                }
            }
        }
        int mid = -1;
        if (best_axis >= 0) {
            float ext = cb.hi[best_axis] - cb.lo[best_axis];
            float scale = NBINS / ext;
            auto it = std::partition(idx.begin() + first, idx.begin() + first + count, [&](int t) {
                int b = (int)((cent[3 * (size_t)t + best_axis] - cb.lo[best_axis]) * scale);
                b = std::min(std::max(b, 0), NBINS - 1);
                return b <= best_split;
            });
            mid = (int)(it - idx.begin());
            int nl = mid - first, nr = count - nl;
            // depth budget: both sides must still fit with median splits
            if (nl == 0 || nr == 0 || depth + 1 + levels_needed(std::max(nl, nr)) > DEPTH_BUDGET) mid = -1;
        }
        if (mid < 0) {
            int axis = 0;
            float e = cb.hi[0] - cb.lo[0];
            for (int k = 1; k < 3; ++k)
                if (cb.hi[k] - cb.lo[k] > e) {
                    e = cb.hi[k] - cb.lo[k];
                    axis = k;
                }
            mid = first + count / 2;
            std::nth_element(idx.begin() + first, idx.begin() + mid, idx.begin() + first + count,
                             [&](int a, int b) { return cent[3 * (size_t)a + axis] < cent[3 * (size_t)b + axis]; });
        }
        int node_index = (int)out.nodes.size();
        out.nodes.emplace_back();
        Box b0, b1;
        int c0 = build(first, mid - first, depth + 1, b0);
        int c1 = build(mid, first + count - mid, depth + 1, b1);
        BVHNode& n = out.nodes[node_index];
        for (int k = 0; k < 3; ++k) {
            n.lo0[k] = b0.lo[k];
            n.hi0[k] = b0.hi[k];
            n.lo1[k] = b1.lo[k];
            n.hi1[k] = b1.hi[k];
        }
        n.c0 = c0;
        n.c1 = c1;
        return node_index;
    }
};
}  // namespace

void build_bvh(const float* positions, int n_tris, BVH& out, int leaf_size, int bins) {
    out.nodes.clear();
    out.leaf_prims.clear();
    out.max_depth = 0;
    out.root_ref = ~0;  // empty leaf marker handled by n_tris == 0
    if (n_tris <= 0) return;
    Builder b(positions, n_tris, out);
    b.LEAF = leaf_size >= 1 && leaf_size <= 8 ? leaf_size : 4;
    b.NBINS = std::min(std::max(bins, 4), (int)Builder::MAXBINS);   // (HK_BVH_BINS)   // (HK_BVH_LEAF, resolved by the caller from its context's knobs)
    Box box;
    out.root_ref = b.build(0, n_tris, 0, box);
    out.max_depth = b.max_depth_seen;
    // Breadth-first renumbering: the recursion above emits nodes in pre-order; the traversal kernels keep the FIRST nodes of the
    // array in LDS, and those should be the top levels of the tree (every ray visits them).  Indices are internal to the node
    // array, so results do not change.
    if (out.root_ref >= 0) {
        const size_t n = out.nodes.size();
        std::vector<int> order;
        order.reserve(n);
        std::vector<int> new_index(n, -1);
        order.push_back(out.root_ref);
        for (size_t head = 0; head < order.size(); ++head) {
            const BVHNode& nd = out.nodes[order[head]];
            if (nd.c0 >= 0) order.push_back(nd.c0);
            if (nd.c1 >= 0) order.push_back(nd.c1);
        }
        for (size_t i = 0; i < order.size(); ++i) new_index[order[i]] = (int)i;
        std::vector<BVHNode> sorted(order.size());
        for (size_t i = 0; i < order.size(); ++i) {
            BVHNode nd = out.nodes[order[i]];
            if (nd.c0 >= 0) nd.c0 = new_index[nd.c0];
            if (nd.c1 >= 0) nd.c1 = new_index[nd.c1];
            sorted[i] = nd;
        }
        out.nodes.swap(sorted);
        out.root_ref = 0;
    }
    for (int k = 0; k < 3; ++k) {
        out.lo[k] = box.lo[k];
        out.hi[k] = box.hi[k];
    }
}

}  // namespace hk

// hko_accel.h — CPU ORACLE (test infrastructure): triangle intersection + a deliberately simple BVH.
//
// PARITY UNPINNED at this boundary: the reference calls Raycore.closest_hit
// (src/integrators/volpath/intersection.jl:200,225,323,703) and Raycore.jl is a third-party
// dependency that is absent from /root/reference (Project.toml:15,33-34, rev "sd/multitype-vec",
// no Manifest).  The contract the call sites rely on is: closest_hit(accel, ray) ->
// (hit, primitive, t, barycentric) with barycentric = (w,u,v) weights of (v0,v1,v2)
// (intersection.jl:27-37) and t in (0, ray.t_max).  This file *defines* the arithmetic the build
// uses for it (DESIGN.md "Intersection arithmetic"):
//
//   Moller-Trumbore on (v0, e1=v1-v0, e2=v2-v0), strict binary32, no FMA:
//     p = d x e2; det = e1.p; if det == 0 -> miss; inv = 1/det; s = o - v0
//     u = (s.p)*inv; reject u<0 or u>1;  q = s x e1;  v = (d.q)*inv; reject v<0 or u+v>1
//     t = (e2.q)*inv; accept iff 0 < t < t_max
//   closest = lexicographic min of (t, triangle index); ties on t go to the smaller index.
//
// The acceleration structure is result-neutral (only conservative culling), so the oracle uses its
// own median-split BVH, different from the product's SAH builder on purpose.
#pragma once
#include <algorithm>
#include <vector>

#include "hko_core.h"

namespace hko {

struct Hit {
    bool hit = false;
    int32_t prim = -1;
    float t = INF_F;
    float u = 0, v = 0;  // barycentric weights of v1, v2 (w = 1-u-v of v0)
};

inline bool intersect_triangle(V3 o, V3 d, float t_max, V3 v0, V3 e1, V3 e2, float& t, float& u, float& v) {
    V3 p = cross(d, e2);
    float det = dot(e1, p);
    if (det == 0.0f) return false;
    float inv = 1.0f / det;
    V3 s = o - v0;
    u = dot(s, p) * inv;
    if (!(u >= 0.0f) || u > 1.0f) return false;
    V3 q = cross(s, e1);
    v = dot(d, q) * inv;
    if (!(v >= 0.0f) || u + v > 1.0f) return false;
    t = dot(e2, q) * inv;
    return t > 0.0f && t < t_max;
}

struct Accel {
    struct Node {
        float lo[3], hi[3];
        int32_t left, right;   // children (inner) ; leaf: left = first, right = -count
    };
    std::vector<Node> nodes;
    std::vector<int32_t> order;  // triangle ids in leaf order
    const float* pos = nullptr;  // [T][3][3]
    int32_t n_tris = 0;
    std::vector<V3> v0s, e1s, e2s;

    void build(const float* positions, int32_t n) {
        pos = positions;
        n_tris = n;
        v0s.resize(n);
        e1s.resize(n);
        e2s.resize(n);
        std::vector<V3> cent(n);
        order.resize(n);
        for (int32_t i = 0; i < n; ++i) {
            const float* p = positions + 9 * (size_t)i;
            V3 a(p[0], p[1], p[2]), b(p[3], p[4], p[5]), c(p[6], p[7], p[8]);
            v0s[i] = a;
            e1s[i] = b - a;
            e2s[i] = c - a;
            cent[i] = V3((a.x + b.x + c.x), (a.y + b.y + c.y), (a.z + b.z + c.z));
            order[i] = i;
        }
        nodes.clear();
        nodes.reserve(2 * (size_t)n + 1);
        if (n > 0) build_rec(0, n, cent);
    }
    int32_t build_rec(int32_t first, int32_t last, const std::vector<V3>& cent) {
        Node nd;
        for (int k = 0; k < 3; ++k) {
            nd.lo[k] = INF_F;
            nd.hi[k] = -INF_F;
        }
        float clo[3] = {INF_F, INF_F, INF_F}, chi[3] = {-INF_F, -INF_F, -INF_F};
        for (int32_t i = first; i < last; ++i) {
            const float* p = pos + 9 * (size_t)order[i];
            for (int vtx = 0; vtx < 3; ++vtx)
                for (int k = 0; k < 3; ++k) {
                    nd.lo[k] = std::min(nd.lo[k], p[3 * vtx + k]);
                    nd.hi[k] = std::max(nd.hi[k], p[3 * vtx + k]);
                }
            for (int k = 0; k < 3; ++k) {
                clo[k] = std::min(clo[k], cent[order[i]][k]);
                chi[k] = std::max(chi[k], cent[order[i]][k]);
            }
        }
        int32_t idx = (int32_t)nodes.size();
        nodes.push_back(nd);
        int32_t count = last - first;
        if (count <= 2) {
            nodes[idx].left = first;
            nodes[idx].right = -count;
            return idx;
        }
        int axis = 0;
        float ext = chi[0] - clo[0];
        for (int k = 1; k < 3; ++k)
            if (chi[k] - clo[k] > ext) {
                ext = chi[k] - clo[k];
                axis = k;
            }
        int32_t mid = (first + last) / 2;
        std::nth_element(order.begin() + first, order.begin() + mid, order.begin() + last,
                         [&](int32_t a, int32_t b) { return cent[a][axis] < cent[b][axis]; });
        int32_t l = build_rec(first, mid, cent);
        int32_t r = build_rec(mid, last, cent);
        nodes[idx].left = l;
        nodes[idx].right = r;
        return idx;
    }

    // conservative slab test: returns true when the box may contain a hit with t <= t_best
    static bool box_may_hit(const Node& n, V3 o, V3 inv_d, float t_best) {
        float tmin = 0.0f, tmax = t_best;
        for (int k = 0; k < 3; ++k) {
            float t0 = (n.lo[k] - o[k]) * inv_d[k];
            float t1 = (n.hi[k] - o[k]) * inv_d[k];
            if (t0 != t0 || t1 != t1) continue;  // 0 * inf: origin on a slab of a flat box -> keep
            float a = std::min(t0, t1), b = std::max(t0, t1);
            a = a - std::fabs(a) * 4e-7f - 1e-30f;  // widen: culling must never drop an equal-t hit
            b = b + std::fabs(b) * 4e-7f + 1e-30f;
            tmin = std::max(tmin, a);
            tmax = std::min(tmax, b);
        }
        return tmin <= tmax;
    }

    Hit closest_hit(V3 o, V3 d, float t_max, uint64_t* n_nodes = nullptr, uint64_t* n_tris_tested = nullptr) const {
        Hit h;
        if (nodes.empty()) return h;
        V3 inv_d(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
        float best = t_max;
        int32_t stack[128];
        int sp = 0;
        stack[sp++] = 0;
        while (sp) {
            const Node& n = nodes[stack[--sp]];
            if (n_nodes) ++*n_nodes;
            if (!box_may_hit(n, o, inv_d, h.hit ? h.t : best)) continue;
            if (n.right < 0) {
                for (int32_t i = n.left; i < n.left - n.right; ++i) {
                    int32_t tri = order[i];
                    float t, u, v;
                    if (n_tris_tested) ++*n_tris_tested;
                    if (intersect_triangle(o, d, t_max, v0s[tri], e1s[tri], e2s[tri], t, u, v)) {
                        if (!h.hit || t < h.t || (t == h.t && tri < h.prim)) {
                            h.hit = true;
                            h.t = t;
                            h.prim = tri;
                            h.u = u;
                            h.v = v;
                        }
                    }
                }
            } else {
                stack[sp++] = n.left;
                stack[sp++] = n.right;
            }
        }
        return h;
    }
};

}  // namespace hko

// light_bvh.cpp — host-side light-BVH build for the device light sampler.
//
// Must produce the SAME tree as the reference's CPU build because the sampling pmf depends on the
// tree shape: BVHLightSampler(lights) + _build_bvh! (src/lights/bvh-light-sampler.jl:283-466),
// LightBounds / DirectionCone union (src/lights/light-bounds.jl:24-158), light_bounds per light
// type (:231-295), _evaluate_cost (:242-259).  12 buckets, swap partition, bit trails by depth.
#include <cmath>
#include <cstring>
#include <utility>

#include "bvh_build.h"

namespace hk {
namespace {

const float kPi = 3.14159265358979323846f;
const float kInf = INFINITY;

struct f3 {
    float x, y, z;
};
inline f3 mk(float x, float y, float z) { return f3{x, y, z}; }
inline f3 add(f3 a, f3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
inline f3 sub(f3 a, f3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
inline f3 mul(f3 a, float s) { return mk(a.x * s, a.y * s, a.z * s); }
inline float dt(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline f3 crs(f3 a, f3 b) { return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
inline float len(f3 a) { return std::sqrt(dt(a, a)); }
inline f3 nrm(f3 a) { return mul(a, 1.0f / len(a)); }
inline float cl(float v, float lo, float hi) { return v < lo ? lo : (v > hi ? hi : v); }
inline float mn(float a, float b) { return a < b ? a : b; }
inline float mx(float a, float b) { return a > b ? a : b; }
inline float comp(f3 a, int k) { return k == 0 ? a.x : (k == 1 ? a.y : a.z); }

struct Cone {
    f3 w;
    float c;
};
struct LB {
    f3 lo, hi, w;
    float phi, cos_o, cos_e;
    bool two_sided;
};
LB empty_lb() { return LB{mk(kInf, kInf, kInf), mk(-kInf, -kInf, -kInf), mk(0, 0, 1), 0.0f, 1.0f, 1.0f, false}; }

float angle_between(f3 a, f3 b) {
    if (dt(a, b) < 0.0f) return kPi - 2.0f * std::asin(cl(len(add(a, b)) * 0.5f, -1.0f, 1.0f));
    return 2.0f * std::asin(cl(len(sub(b, a)) * 0.5f, -1.0f, 1.0f));
}
Cone cone_union(Cone a, Cone b) {
    if (a.c == kInf) return b;
    if (b.c == kInf) return a;
    float ta = std::acos(cl(a.c, -1.0f, 1.0f)), tb = std::acos(cl(b.c, -1.0f, 1.0f));
    float td = angle_between(a.w, b.w);
    if (mn(td + tb, kPi) <= ta) return a;
    if (mn(td + ta, kPi) <= tb) return b;
    float to = (ta + td + tb) * 0.5f;
    Cone sphere{mk(0, 0, 1), -1.0f};
    if (to >= kPi) return sphere;
    float tr = to - ta;
    f3 wr = crs(a.w, b.w);
    if (dt(wr, wr) == 0.0f) return sphere;
    f3 axis = nrm(wr);
    float s = std::sin(tr), c = std::cos(tr);
    f3 w = add(add(mul(a.w, c), mul(crs(axis, a.w), s)), mul(mul(axis, dt(axis, a.w)), 1.0f - c));
    return Cone{nrm(w), std::cos(to)};
}
LB lb_union(const LB& a, const LB& b) {
    if (a.phi == 0.0f) return b;
    if (b.phi == 0.0f) return a;
    Cone c = cone_union(Cone{a.w, a.cos_o}, Cone{b.w, b.cos_o});
    LB r;
    r.lo = mk(mn(a.lo.x, b.lo.x), mn(a.lo.y, b.lo.y), mn(a.lo.z, b.lo.z));
    r.hi = mk(mx(a.hi.x, b.hi.x), mx(a.hi.y, b.hi.y), mx(a.hi.z, b.hi.z));
    r.w = c.w;
    r.phi = a.phi + b.phi;
    r.cos_o = c.c;
    r.cos_e = mn(a.cos_e, b.cos_e);
    r.two_sided = a.two_sided || b.two_sided;
    return r;
}
f3 centroid(const LB& l) { return mul(add(l.lo, l.hi), 0.5f); }

float poly_sigmoid(float x) {
    if (std::isinf(x)) return x > 0 ? 1.0f : 0.0f;
    return 0.5f + x / (2.0f * std::sqrt(1.0f + x * x));
}
}  // namespace

float poly_eval(const float c[3], float lambda) { return poly_sigmoid(c[0] * lambda * lambda + c[1] * lambda + c[2]); }
float poly_max(const float c[3]) {
    float r = mx(poly_eval(c, 360.0f), poly_eval(c, 830.0f));
    if (c[0] != 0) {
        float lc = -c[1] / (2.0f * c[0]);
        if (360.0f <= lc && lc <= 830.0f) r = mx(r, poly_eval(c, lc));
    }
    return r;
}

namespace {
float spectrum_luminance(const hk_light& l) {
    if (l.spectrum_kind == HK_SPEC_ILLUMINANT) return l.illum_scale * poly_max(l.poly) * 100.0f;  // max_value(::RGBIlluminantSpectrum)
    return 0.212671f * l.i_rgb[0] + 0.715160f * l.i_rgb[1] + 0.072169f * l.i_rgb[2];
}
bool bounds_of(const hk_light& l, LB& o) {
    const float cos_pi = (float)std::cos(3.14159265358979323846);
    const float cos_half_pi = (float)std::cos(3.14159265358979323846 / 2);
    o = empty_lb();
    if (l.kind == HK_LIGHT_POINT || l.kind == HK_LIGHT_SPOT) {
        f3 p = mk(l.position[0], l.position[1], l.position[2]);
        o.lo = o.hi = p;
        o.phi = 4.0f * kPi * l.scale * spectrum_luminance(l);
        o.two_sided = false;
        if (l.kind == HK_LIGHT_POINT) {
            o.w = mk(0, 0, 1);
            o.cos_o = cos_pi;
            o.cos_e = cos_half_pi;
        } else {
            o.w = nrm(mk(l.light_to_world[2], l.light_to_world[6], l.light_to_world[10]));
            float ce = (float)std::cos(std::acos(l.cos_total_width) - std::acos(l.cos_falloff_start));
            if (ce == 1.0f && l.cos_total_width != l.cos_falloff_start) ce = 0.999f;
            o.cos_o = l.cos_falloff_start;
            o.cos_e = ce;
        }
        return true;
    }
    if (l.kind == HK_LIGHT_DIFFUSE_AREA) {
        for (int k = 0; k < 3; ++k) {
            f3 v = mk(l.v[3 * k], l.v[3 * k + 1], l.v[3 * k + 2]);
            if (k == 0)
                o.lo = o.hi = v;
            else {
                o.lo = mk(mn(o.lo.x, v.x), mn(o.lo.y, v.y), mn(o.lo.z, v.z));
                o.hi = mk(mx(o.hi.x, v.x), mx(o.hi.y, v.y), mx(o.hi.z, v.z));
            }
        }
        float sided = l.two_sided ? 2.0f : 1.0f;
        float lum = l.Le.tex < 0 ? (0.212671f * l.Le.c[0] + 0.715160f * l.Le.c[1] + 0.072169f * l.Le.c[2]) : l.scale;
        o.w = mk(l.normal[0], l.normal[1], l.normal[2]);
        o.phi = kPi * sided * l.area * l.scale * lum;
        o.cos_o = 1.0f;
        o.cos_e = cos_half_pi;
        o.two_sided = l.two_sided != 0;
        return true;
    }
    return false;  // directional / sun / ambient / environment: infinite
}

float split_cost(const LB& lb, f3 blo, f3 bhi, int dim) {
    float to = std::acos(cl(lb.cos_o, -1.0f, 1.0f)), te = std::acos(cl(lb.cos_e, -1.0f, 1.0f));
    float tw = mn(to + te, kPi);
    float so = std::sqrt(mx(0.0f, 1.0f - lb.cos_o * lb.cos_o));
    float M = 2.0f * kPi * (1.0f - lb.cos_o) + kPi / 2.0f * (2.0f * tw * so - std::cos(to - 2.0f * tw) - 2.0f * to * so + lb.cos_o);
    f3 d = sub(bhi, blo);
    float maxd = mx(mx(d.x, d.y), d.z), dd = comp(d, dim);
    float Kr = dd > 1e-10f ? maxd / dd : maxd / 1e-10f;
    float area = 2.0f * (d.x * d.y + d.x * d.z + d.y * d.z);
    return lb.phi * M * Kr * area;
}
int bucket(f3 clo, f3 chi, f3 c, int dim) {
    float o = comp(c, dim) - comp(clo, dim);
    if (comp(chi, dim) > comp(clo, dim)) o /= (comp(chi, dim) - comp(clo, dim));
    int b = (int)std::floor(12 * o);
    return (b < 0 ? 0 : (b > 11 ? 11 : b)) + 1;
}

struct Rec {
    std::vector<std::pair<int, LB>>& L;
    LightBVH& out;
    LightBVHNodeH node_of(const LB& lb, uint32_t child, bool leaf) {
        LightBVHNodeH n;
        std::memset(&n, 0, sizeof n);
        n.bmin[0] = lb.lo.x; n.bmin[1] = lb.lo.y; n.bmin[2] = lb.lo.z;
        n.bmax[0] = lb.hi.x; n.bmax[1] = lb.hi.y; n.bmax[2] = lb.hi.z;
        n.w[0] = lb.w.x; n.w[1] = lb.w.y; n.w[2] = lb.w.z;
        n.phi = lb.phi;
        n.cos_o = lb.cos_o;
        n.cos_e = lb.cos_e;
        n.bits = (lb.two_sided ? 1u : 0u) | (leaf ? 2u : 0u);
        n.child1_or_light = child;
        return n;
    }
    LB go(int start, int stop, uint32_t trail, int depth) {  // 1-based inclusive
        int count = stop - start + 1;
        if (count == 1) {
            auto& it = L[start - 1];
            out.nodes.push_back(node_of(it.second, (uint32_t)it.first, true));
            out.bit_trails[it.first - 1] = trail;
            return it.second;
        }
        LB all = L[start - 1].second;
        f3 clo = centroid(all), chi = clo;
        for (int i = start + 1; i <= stop; ++i) {
            all = lb_union(all, L[i - 1].second);
            f3 c = centroid(L[i - 1].second);
            clo = mk(mn(clo.x, c.x), mn(clo.y, c.y), mn(clo.z, c.z));
            chi = mk(mx(chi.x, c.x), mx(chi.y, c.y), mx(chi.z, c.z));
        }
        float best = kInf;
        int bdim = 0, bbucket = 0;
        for (int dim = 1; dim <= 3; ++dim) {
            if (!(comp(chi, dim - 1) - comp(clo, dim - 1) > 0.0f)) continue;
            LB bb[12];
            int bc[12];
            for (int k = 0; k < 12; ++k) {
                bb[k] = empty_lb();
                bc[k] = 0;
            }
            for (int i = start; i <= stop; ++i) {
                int b = bucket(clo, chi, centroid(L[i - 1].second), dim - 1);
                bb[b - 1] = lb_union(bb[b - 1], L[i - 1].second);
                bc[b - 1]++;
            }
            for (int split = 1; split <= 11; ++split) {
                LB lo = empty_lb(), hi = empty_lb();
                int nlo = 0, nhi = 0;
                for (int b = 1; b <= split; ++b) {
                    lo = lb_union(lo, bb[b - 1]);
                    nlo += bc[b - 1];
                }
                for (int b = split + 1; b <= 12; ++b) {
                    hi = lb_union(hi, bb[b - 1]);
                    nhi += bc[b - 1];
                }
                if (nlo == 0 || nhi == 0) continue;
                float cost = split_cost(lo, all.lo, all.hi, dim - 1) + split_cost(hi, all.lo, all.hi, dim - 1);
                if (cost < best) {
                    best = cost;
                    bdim = dim;
                    bbucket = split;
                }
            }
        }
        int mid;
        if (bdim > 0) {
            int pivot = start;
            for (int i = start; i <= stop; ++i) {
                if (bucket(clo, chi, centroid(L[i - 1].second), bdim - 1) <= bbucket) {
                    if (i != pivot) std::swap(L[pivot - 1], L[i - 1]);
                    ++pivot;
                }
            }
            mid = (pivot == start || pivot > stop) ? start + count / 2 : pivot - 1;
        } else
            mid = start + count / 2 - 1;
        if (mid < start) mid = start;
        if (mid > stop - 1) mid = stop - 1;
        size_t me = out.nodes.size();
        out.nodes.push_back(node_of(all, 0, false));
        LB l0 = go(start, mid, trail, depth + 1);
        uint32_t child1 = (uint32_t)out.nodes.size() + 1;
        LB l1 = go(mid + 1, stop, trail | (depth >= 32 ? 0u : (1u << depth)), depth + 1);
        out.nodes[me] = node_of(lb_union(l0, l1), child1, false);
        return lb_union(l0, l1);
    }
};
}  // namespace

void build_light_bvh(const hk_light* lights, int n, LightBVH& out) {
    out.nodes.clear();
    out.infinite.clear();
    out.bit_trails.assign((size_t)n, 0xFFFFFFFFu);
    std::vector<std::pair<int, LB>> L;
    for (int i = 1; i <= n; ++i) {
        LB lb;
        if (!bounds_of(lights[i - 1], lb))
            out.infinite.push_back(i);
        else if (lb.phi > 0.0f)
            L.emplace_back(i, lb);
    }
    out.num_bvh = (int)L.size();
    if (!L.empty()) {
        Rec r{L, out};
        r.go(1, (int)L.size(), 0u, 0);
    }
}

// rgb_to_spectrum (spectral/rgb2spec.jl:85-167): host-side bake of constant colours
void rgb_to_coeffs(const RGB2Spec& t, float r, float g, float b, float out[3]) {
    r = cl(r, 0.0f, 1.0f);
    g = cl(g, 0.0f, 1.0f);
    b = cl(b, 0.0f, 1.0f);
    if (r == g && g == b) {
        out[0] = out[1] = 0.0f;
        out[2] = (r > 0.0f && r < 1.0f) ? (r - 0.5f) / std::sqrt(r * (1.0f - r)) : (r <= 0.0f ? -1.0e10f : 1.0e10f);
        return;
    }
    int maxc = r > g ? (r > b ? 1 : 3) : (g > b ? 2 : 3);
    float z = maxc == 1 ? r : (maxc == 2 ? g : b);
    float xc = maxc == 1 ? g : (maxc == 2 ? b : r);
    float yc = maxc == 1 ? b : (maxc == 2 ? r : g);
    int res = t.res;
    float x = xc * (float)(res - 1) / z, y = yc * (float)(res - 1) / z;
    int zi = 1;
    for (int i = 1; i <= res - 1; ++i)
        if (t.scale[i - 1] < z) zi = i;
    if (zi > res - 1) zi = res - 1;
    int xi = (int)x + 1, yi = (int)y + 1;
    if (xi > res - 1) xi = res - 1;
    if (yi > res - 1) yi = res - 1;
    float dx = x - (float)(xi - 1), dy = y - (float)(yi - 1);
    float dz = (z - t.scale[zi - 1]) / (t.scale[zi] - t.scale[zi - 1]);
    size_t R = (size_t)res;
    auto at = [&](int zi_, int yi_, int xi_, int k) {
        return t.coeffs[(size_t)(maxc - 1) + 3 * ((size_t)(zi_ - 1) + R * ((size_t)(yi_ - 1) + R * ((size_t)(xi_ - 1) + R * (size_t)(k - 1))))];
    };
    for (int k = 1; k <= 3; ++k) {
        out[k - 1] = (1.0f - dz) * ((1.0f - dy) * ((1.0f - dx) * at(zi, yi, xi, k) + dx * at(zi, yi, xi + 1, k)) +
                                    dy * ((1.0f - dx) * at(zi, yi + 1, xi, k) + dx * at(zi, yi + 1, xi + 1, k))) +
                     dz * ((1.0f - dy) * ((1.0f - dx) * at(zi + 1, yi, xi, k) + dx * at(zi + 1, yi, xi + 1, k)) +
                           dy * ((1.0f - dx) * at(zi + 1, yi + 1, xi, k) + dx * at(zi + 1, yi + 1, xi + 1, k)));
    }
}

}  // namespace hk

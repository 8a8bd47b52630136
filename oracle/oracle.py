"""ctypes wrapper around oracle/liboracle.so — the CPU ORACLE (test infrastructure).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product
package (hikari.jl_amd) never does."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liboracle.so")
_lib = None


def build(force=False):
    if force or not os.path.isfile(LIB_PATH):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        import hikari_jl_amd as hk
        A = hk._abi
        L = C.CDLL(LIB_PATH)
        vp, i32, PF, PI = C.c_void_p, C.c_int32, A.PF, C.POINTER(C.c_int32)
        L.hko_scene_create.argtypes = [C.POINTER(A.hk_scene_desc), C.POINTER(A.hk_tables), C.POINTER(vp)]
        L.hko_scene_destroy.argtypes = [vp]
        L.hko_render.argtypes = [vp, C.POINTER(A.hk_integrator_params), C.POINTER(A.hk_camera), i32, i32, i32, i32, i32, vp, C.POINTER(A.hk_stats)]
        L.hko_finalize.argtypes = [i32, i32, i32, vp, PF]
        L.hko_trace_closest.argtypes = [vp, i32, PF, PF, PF, PF, PI, PF]
        L.hko_sobol.argtypes = [C.POINTER(C.c_uint32), i32, i32, i32, C.c_uint32, i32, PI, PI, PI, PI, PF, PF]
        L.hko_camera.argtypes = [C.POINTER(C.c_uint32), C.POINTER(A.hk_integrator_params), C.POINTER(A.hk_camera), i32, i32, i32, PI, PI, PI, PF]
        L.hko_uplift.argtypes = [C.POINTER(A.hk_tables), i32, i32, PF, PF, PF]
        L.hko_light_bvh.argtypes = [vp, i32, PF, PF, PF, PI, PF, PI, PF]
        L.hko_fill_aux.argtypes = [vp, C.POINTER(A.hk_camera), i32, i32, i32, PF, PF, PF]
        L.hko_postprocess.argtypes = [C.POINTER(A.hk_postprocess_params), i32, i32, PF, PF, PF]
        L.hko_denoise.argtypes = [C.POINTER(A.hk_denoise_params), i32, i32, PF, PF, PF, PF, PF]
        L.hko_light.argtypes = [vp, i32, i32, i32, PF, PF, PF, PF]
        L.hko_bsdf.argtypes = [vp, i32, i32, i32, i32, PF, PF, PF, PF, PF, PF, PF]
        L.hko_light_bvh_copy.argtypes = [vp, PI, PF, C.POINTER(C.c_uint32)]
        L.hko_mix_resolve.argtypes = [vp, i32, i32, PF, PF, PF, PI]
        L.hko_medium.argtypes = [vp, i32, i32, i32, PF, PF, PF, PF, PF]
        L.hko_murmur64a.argtypes = [C.c_char_p, i32, C.c_uint64]
        L.hko_murmur64a.restype = C.c_uint64
        L.hko_mix_bits.argtypes = [C.c_uint64]
        L.hko_mix_bits.restype = C.c_uint64
        L.hko_pcg32.argtypes = [C.c_uint64, C.c_uint64, i32, i32, C.POINTER(C.c_uint32), PF]
        L.hko_pcg32.restype = None
        for n in ("hko_fresnel_dielectric", "hko_fr_complex", "hko_sample_d65"):
            getattr(L, n).restype = C.c_float
        L.hko_fresnel_dielectric.argtypes = [C.c_float, C.c_float]
        L.hko_fr_complex.argtypes = [C.c_float, C.c_float, C.c_float]
        L.hko_sample_d65.argtypes = [C.c_float]
        L.hko_filter_eval.argtypes = [C.POINTER(A.hk_integrator_params), C.c_float, C.c_float]
        L.hko_filter_eval.restype = C.c_float
        L.hko_filter_sample.argtypes = [C.POINTER(A.hk_integrator_params), i32, PF, PF, PF]
        L.hko_filter_sample.restype = None
        L.hko_wavelengths.argtypes = [i32, PF, PF]
        L.hko_wavelengths.restype = None
        L.hko_tr.argtypes = [i32, PF, PF, PF, C.c_float, C.c_float, PF]
        L.hko_tr.restype = None
        L.hko_hg.argtypes = [i32, C.c_float, PF, PF, PF, PF]
        L.hko_hg.restype = None
        L.hko_equal_area.argtypes = [i32, PF, PF, PF]
        L.hko_equal_area.restype = None
        L.hko_dist2d.argtypes = [C.POINTER(A.hk_envmap), i32, PF, PF, PF]
        L.hko_dist2d.restype = None
        L.hko_node_importance.argtypes = [vp, i32, i32, PF, PF, PF]
        L.hko_node_importance.restype = None
        L.hko_cosine_hemisphere.argtypes = [i32, PF, PF]
        L.hko_cosine_hemisphere.restype = None
        L.hko_set_threads.argtypes = [i32]
        L.hko_set_threads.restype = None
        L.hko_max_threads.restype = i32
        _lib = L
        set_threads(min(os.cpu_count() or 1, 16))   # tests render tiny frames; bench raises this to all cores
    return _lib


def set_threads(n):
    lib().hko_set_threads(int(n))


def available_cores():
    """Host cores this process may actually use: min(affinity mask, cgroup CPU quota).  A container with a 16-CPU quota on a
    256-thread host runs 256 oracle threads 10x slower than 16."""
    import math
    import os
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]          # cgroup v2
        if quota != "max":
            n = min(n, max(1, math.ceil(int(quota) / int(period))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())          # cgroup v1
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0 and per > 0:
                n = min(n, max(1, math.ceil(q / per)))
        except (OSError, ValueError):
            pass
    return max(n, 1)


def max_threads():
    return int(lib().hko_max_threads())


def _pf(a):
    import hikari_jl_amd as hk
    return a.ctypes.data_as(hk._abi.PF)


def _pi(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


class OracleScene:
    def __init__(self, scene):
        import hikari_jl_amd as hk
        self.scene = scene  # keeps the borrowed arrays alive
        self.tables = hk.tables.load()
        self.h = C.c_void_p()
        st = lib().hko_scene_create(C.byref(scene.desc), C.byref(self.tables["struct"]), C.byref(self.h))
        if st != 0:
            raise RuntimeError("hko_scene_create failed: %d" % st)

    def render(self, params, camera, width, height, n_samples, first=1, stride=1, accum=None):
        """-> (accum[4N], stats).  accum = [rgb 3N | weight N]."""
        import hikari_jl_amd as hk
        n = width * height
        dt = np.float64 if params.accumulate_f64 else np.float32
        if accum is None:
            accum = np.zeros(4 * n, dtype=dt)
        stats = hk._abi.hk_stats()
        cam = camera.record()
        st = lib().hko_render(self.h, C.byref(params), C.byref(cam), width, height, first, n_samples, stride,
                              accum.ctypes.data_as(C.c_void_p), C.byref(stats))
        assert st == 0
        return accum, stats

    def trace(self, o, d, tmax):
        n = o.shape[0]
        o = np.ascontiguousarray(o, np.float32)
        d = np.ascontiguousarray(d, np.float32)
        tmax = np.ascontiguousarray(tmax, np.float32)
        t = np.empty(n, np.float32)
        prim = np.empty(n, np.int32)
        uv = np.empty((n, 2), np.float32)
        lib().hko_trace_closest(self.h, n, _pf(o), _pf(d), _pf(tmax), _pf(t), _pi(prim), _pf(uv))
        return t, prim, uv

    def light_bvh(self, p, n, u, query=None):
        m = p.shape[0]
        p = np.ascontiguousarray(p, np.float32)
        n = np.ascontiguousarray(n, np.float32)
        u = np.ascontiguousarray(u, np.float32)
        li = np.empty(m, np.int32)
        pmf = np.empty(m, np.float32)
        qp = np.zeros(m, np.float32)
        q = np.ascontiguousarray(query if query is not None else np.zeros(m), np.int32)
        lib().hko_light_bvh(self.h, m, _pf(p), _pf(n), _pf(u), _pi(li), _pf(pmf), _pi(q), _pf(qp))
        return li, pmf, qp

    def fill_aux(self, camera, width, height, has_infinite_lights=False):
        """fill_aux_buffers! -> (albedo [h,w,3], normal [h,w,3], depth [h,w])"""
        n = width * height
        a, nn, d = np.zeros((width, height, 3), np.float32), np.zeros((width, height, 3), np.float32), np.zeros((width, height), np.float32)
        cam = camera.record()
        lib().hko_fill_aux(self.h, C.byref(cam), width, height, 1 if has_infinite_lights else 0, _pf(a), _pf(nn), _pf(d))
        return np.transpose(a, (1, 0, 2)).copy(), np.transpose(nn, (1, 0, 2)).copy(), np.transpose(d, (1, 0)).copy()

    def light(self, mode, light_idx_1based, p, x, lam):
        """mode 0: sample_light_spectral(light, p, lambda, u = x[:, :2]) -> [n, 12] = wi3, pdf, Li4, p_light3, is_delta;
        mode 1: escaped ray along x -> Le4 (all lights), env pdf"""
        a = [np.ascontiguousarray(v, np.float32) for v in (p, x, lam)]
        n = a[0].shape[0]
        out = np.zeros((n, 12), np.float32)
        lib().hko_light(self.h, mode, light_idx_1based, n, *[_pf(v) for v in a], _pf(out))
        return out

    def bsdf(self, mode, mat_idx, wo, wi, ns, lam, u, uc, regularize=False):
        """point-wise sample_bsdf_spectral (mode 0) / evaluate_bsdf_spectral (mode 1) -> [n, 10]"""
        a = [np.ascontiguousarray(x, np.float32) for x in (wo, wi, ns, lam, u, uc)]
        n = a[0].shape[0]
        out = np.zeros((n, 10), np.float32)
        lib().hko_bsdf(self.h, mode, mat_idx, 1 if regularize else 0, n, *[_pf(x) for x in a], _pf(out))
        return out

    def mix_resolve(self, mat_idx, p, wo, uv):
        """resolve_mix_material(mat_idx, p, wo, uv) -> material index per point"""
        a = [np.ascontiguousarray(x, np.float32) for x in (p, wo, uv)]
        out = np.empty(a[0].shape[0], np.int32)
        lib().hko_mix_resolve(self.h, mat_idx, a[0].shape[0], *[_pf(x) for x in a], _pi(out))
        return out

    def medium(self, mode, medium_idx, a, lam, b=None, tmax=None):
        """mode 0: sample_point(medium, p = a, lambda) -> [n, 13] = sigma_a4, sigma_s4, Le4, g;
        mode 1: majorant segments along (o = a, d = b, tmax) -> [n, 49] = count, (t_min, t_max, sigma_maj[1]) x 16;
        mode 2: the shadow walk of the ray (o = a, d = b, tmax) that starts in medium `medium_idx` (-1: vacuum) -> [n, 13] = T_ray4, r_u4, r_l4, visible"""
        n = a.shape[0]
        a = np.ascontiguousarray(a, np.float32)
        lam = np.ascontiguousarray(lam, np.float32)
        b = np.ascontiguousarray(b if b is not None else np.zeros((n, 3)), np.float32)
        tmax = np.ascontiguousarray(tmax if tmax is not None else np.zeros(n), np.float32)
        out = np.zeros((n, 49 if mode == 1 else 13), np.float32)
        lib().hko_medium(self.h, mode, medium_idx, n, _pf(a), _pf(b), _pf(tmax), _pf(lam), _pf(out))
        return out

    def media_stage(self, medium_idx, depth, max_depth, rays23, direct_uc, direct_u, indirect_u):
        """K4 + K5 + K6 of caller-supplied rays through the oracle's own stage code -> [n, 56] (hko_media_stage in hko_render.cpp)"""
        rays23 = np.ascontiguousarray(rays23, np.float32)
        n = rays23.shape[0]
        a = [np.ascontiguousarray(x, np.float32) for x in (direct_uc, direct_u, indirect_u)]
        out = np.zeros((n, 56), np.float32)
        L = lib()
        L.hko_media_stage.restype = C.c_int32
        PF = C.POINTER(C.c_float)
        L.hko_media_stage.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, PF, PF, PF, PF, PF]
        pf = lambda x: x.ctypes.data_as(PF)     # noqa: E731
        assert L.hko_media_stage(self.h, medium_idx, depth, max_depth, n, pf(rays23), *[pf(x) for x in a], pf(out)) == 0
        return out

    def node_importance(self, node_idx, p, n):
        p, n = np.ascontiguousarray(p, np.float32), np.ascontiguousarray(n, np.float32)
        out = np.zeros(p.shape[0], np.float32)
        lib().hko_node_importance(self.h, node_idx, p.shape[0], _pf(p), _pf(n), _pf(out))
        return out

    def light_bvh_nodes(self):
        nn = C.c_int32()
        lib().hko_light_bvh_copy(self.h, C.byref(nn), None, None)
        nodes = np.zeros((max(nn.value, 1), 16), np.float32)
        trails = np.zeros(max(self.scene.desc.n_lights, 1), np.uint32)
        lib().hko_light_bvh_copy(self.h, C.byref(nn), _pf(nodes), trails.ctypes.data_as(C.POINTER(C.c_uint32)))
        return nodes[:nn.value], trails[:self.scene.desc.n_lights]

    def close(self):
        if self.h:
            lib().hko_scene_destroy(self.h)
            self.h = None


def finalize(accum, width, height):
    """K13 -> framebuffer [h, w, 3]"""
    out = np.empty((width, height, 3), np.float32)
    lib().hko_finalize(width, height, 1 if accum.dtype == np.float64 else 0, accum.ctypes.data_as(C.c_void_p), _pf(out))
    return np.transpose(out, (1, 0, 2)).copy()


def sobol(width, height, spp, seed, px, py, sidx, dim):
    import hikari_jl_amd as hk
    t = hk.tables.load()
    n = len(px)
    a = [np.ascontiguousarray(x, np.int32) for x in (px, py, sidx, dim)]
    o1 = np.empty(n, np.float32)
    o2 = np.empty((n, 2), np.float32)
    lib().hko_sobol(t["struct"].sobol_matrices, width, height, spp, seed, n, _pi(a[0]), _pi(a[1]), _pi(a[2]), _pi(a[3]), _pf(o1), _pf(o2))
    return o1, o2


def camera_samples(params, camera, width, height, px, py, sidx):
    import hikari_jl_amd as hk
    t = hk.tables.load()
    n = len(px)
    a = [np.ascontiguousarray(x, np.int32) for x in (px, py, sidx)]
    out = np.empty((n, 15), np.float32)
    cam = camera.record()
    lib().hko_camera(t["struct"].sobol_matrices, C.byref(params), C.byref(cam), width, height, n, _pi(a[0]), _pi(a[1]), _pi(a[2]), _pf(out))
    return out


def uplift(mode, rgb, lam):
    import hikari_jl_amd as hk
    t = hk.tables.load()
    rgb = np.ascontiguousarray(rgb, np.float32)
    lam = np.ascontiguousarray(lam, np.float32)
    out = np.empty_like(lam)
    lib().hko_uplift(C.byref(t["struct"]), mode, rgb.shape[0], _pf(rgb), _pf(lam), _pf(out))
    return out


def denoise(params, framebuffer, normal, depth):
    """denoise! on framebuffer / film.normal [h, w, 3] and film.depth [h, w] -> (film.postprocess, framebuffer after the call)"""
    h, w = framebuffer.shape[:2]
    src = np.ascontiguousarray(np.transpose(framebuffer, (1, 0, 2)), np.float32)       # Julia [h,w] column-major
    nn = np.ascontiguousarray(np.transpose(normal, (1, 0, 2)), np.float32)
    dp = np.ascontiguousarray(np.transpose(depth, (1, 0)), np.float32)
    dst, after = np.empty_like(src), np.empty_like(src)
    lib().hko_denoise(C.byref(params), w, h, _pf(src), _pf(nn), _pf(dp), _pf(dst), _pf(after))
    return np.transpose(dst, (1, 0, 2)).copy(), np.transpose(after, (1, 0, 2)).copy()


def postprocess(params, framebuffer, depth=None):
    """postprocess_kernel! on a framebuffer [h, w, 3] (and film.depth [h, w]) -> [h, w, 3]"""
    h, w = framebuffer.shape[:2]
    src = np.ascontiguousarray(np.transpose(framebuffer, (1, 0, 2)), np.float32)       # Julia [h,w] column-major
    dst = np.empty_like(src)
    dp = np.ascontiguousarray(np.transpose(depth, (1, 0)), np.float32) if depth is not None else None
    lib().hko_postprocess(C.byref(params), w, h, _pf(src), _pf(dp) if dp is not None else None, _pf(dst))
    return np.transpose(dst, (1, 0, 2)).copy()


# ---- point-wise helpers (independent float64 pins, tests/test_independent_pins.py) ----
def _f(a):
    return np.ascontiguousarray(a, np.float32)


def tr(w, wm, u, ax, ay):
    """-> [n, 8] = D(wm), Lambda(w), G1(w), G(w, wm), pdf(w, wm), sample_wm(w, u).xyz"""
    w, wm, u = _f(w), _f(wm), _f(u)
    out = np.zeros((w.shape[0], 8), np.float32)
    lib().hko_tr(w.shape[0], _pf(w), _pf(wm), _pf(u), ax, ay, _pf(out))
    return out


def hg(g, wo, u, cos_in):
    """-> [n, 5] = sample_hg(g, wo, u) (wi.xyz, pdf), hg_p(g, cos_in)"""
    wo, u, c = _f(wo), _f(u), _f(cos_in)
    out = np.zeros((wo.shape[0], 5), np.float32)
    lib().hko_hg(wo.shape[0], g, _pf(wo), _pf(u), _pf(c), _pf(out))
    return out


def equal_area(uv, d):
    """-> [n, 5] = equal_area_square_to_sphere(uv).xyz, equal_area_sphere_to_square(d).uv"""
    uv, d = _f(uv), _f(d)
    out = np.zeros((uv.shape[0], 5), np.float32)
    lib().hko_equal_area(uv.shape[0], _pf(uv), _pf(d), _pf(out))
    return out


def dist2d(envmap_record, u, uv_in):
    """-> [n, 4] = Distribution2D.sample_continuous(u) (uv.xy, pdf), pdf(uv_in)"""
    u, q = _f(u), _f(uv_in)
    out = np.zeros((u.shape[0], 4), np.float32)
    lib().hko_dist2d(C.byref(envmap_record), u.shape[0], _pf(u), _pf(q), _pf(out))
    return out


def cosine_hemisphere(u):
    u = _f(u)
    out = np.zeros((u.shape[0], 3), np.float32)
    lib().hko_cosine_hemisphere(u.shape[0], _pf(u), _pf(out))
    return out

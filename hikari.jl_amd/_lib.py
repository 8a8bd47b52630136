"""Loader for the HIP product library (csrc/libhikari_mi355x.so).  There is NO CPU fallback: if the
extension is missing or fails to load, every entry point of the package raises."""
import ctypes as C
import os

from . import _abi as A

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("HK_LIB_PATH") or os.path.join(_HERE, "csrc", "libhikari_mi355x.so")   # HK_LIB_PATH: A/B builds of the same library (tools/)
_lib = None


class HikariMI355XError(RuntimeError):
    pass


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise HikariMI355XError(
            "HIP extension %s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback." % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    vp = C.c_void_p
    i32 = C.c_int32
    PF = A.PF
    PI = C.POINTER(C.c_int32)
    sig = {
        "hk_ctx_create": ([i32, vp, C.POINTER(vp)], i32),
        "hk_ctx_destroy": ([vp], i32),
        "hk_last_error": ([], C.c_char_p),
        "hk_ctx_set_tables": ([vp, C.POINTER(A.hk_tables)], i32),
        "hk_ctx_set_option": ([vp, C.c_char_p, C.c_char_p], i32),
        "hk_ctx_get_option": ([vp, C.c_char_p, C.c_char_p, i32], i32),
        "hk_trim_cache": ([vp], i32),
        "hk_flush": ([vp], i32),
        "hk_film_read_rgb_async": ([vp, vp], i32),
        "hk_film_read_wait": ([vp, vp, PF, C.POINTER(PF)], i32),
        "hk_film_pin_host": ([vp, PF], i32),
        "hk_film_unpin_host": ([vp], i32),
        "hk_scene_create": ([vp, C.POINTER(A.hk_scene_desc), C.POINTER(vp)], i32),
        "hk_scene_destroy": ([vp], i32),
        "hk_integrator_create": ([vp, C.POINTER(A.hk_integrator_params), C.POINTER(vp)], i32),
        "hk_integrator_destroy": ([vp], i32),
        "hk_film_create": ([vp, i32, i32, i32, vp, C.POINTER(vp)], i32),
        "hk_film_destroy": ([vp], i32),
        "hk_film_clear": ([vp], i32),
        "hk_render": ([vp, vp, vp, vp, C.POINTER(A.hk_camera), i32, i32, i32], i32),
        "hk_render_tile": ([vp, vp, vp, vp, C.POINTER(A.hk_camera), i32, i32, i32, i32, i32, i32, i32], i32),
        "hk_comm_create": ([C.POINTER(vp), i32, C.POINTER(vp)], i32),
        "hk_comm_unique_id": ([C.POINTER(C.c_uint8)], i32),
        "hk_comm_create_rank": ([vp, C.POINTER(C.c_uint8), i32, i32, C.POINTER(vp)], i32),
        "hk_comm_destroy": ([vp], i32),
        "hk_film_reduce": ([vp, C.POINTER(vp), i32, i32], i32),
        "hk_film_read_rgb": ([vp, vp, PF], i32),
        "hk_film_read_accum": ([vp, vp, vp], i32),
        "hk_film_accum_device_ptr": ([vp], vp),
        "hk_sync": ([vp], i32),
        "hk_stats_get": ([vp, C.POINTER(A.hk_stats)], i32),
        "hk_stats_reset": ([vp], i32),
        "hk_stats_enable_counters": ([vp, i32], i32),
        "hk_trace_closest": ([vp, vp, i32, PF, PF, PF, PF, PI, PF], i32),
        "hk_test_sobol": ([vp, i32, i32, i32, C.c_uint32, i32, PI, PI, PI, PI, PF, PF], i32),
        "hk_test_camera": ([vp, vp, C.POINTER(A.hk_camera), i32, i32, i32, PI, PI, PI, PF], i32),
        "hk_test_uplift": ([vp, i32, i32, PF, PF, PF], i32),
        "hk_test_light_bvh": ([vp, vp, i32, PF, PF, PF, PI, PF, PI, PF], i32),
        "hk_film_postprocess": ([vp, vp, C.POINTER(A.hk_postprocess_params), PF, PF], i32),
        "hk_postprocess": ([vp, C.POINTER(A.hk_postprocess_params), i32, i32, PF, PF, PF], i32),
        "hk_denoise": ([vp, C.POINTER(A.hk_denoise_params), i32, i32, PF, PF, PF, PF, PF], i32),
        "hk_film_fill_aux": ([vp, vp, C.POINTER(A.hk_camera), i32, i32, i32, PF, PF, PF], i32),
        "hk_test_light": ([vp, vp, i32, i32, i32, PF, PF, PF, PF], i32),
        "hk_test_bsdf": ([vp, vp, i32, i32, i32, i32, PF, PF, PF, PF, PF, PF, PF], i32),
        "hk_test_mix": ([vp, vp, i32, i32, PF, PF, PF, PI], i32),
        "hk_test_medium": ([vp, vp, i32, i32, i32, PF, PF, PF, PF, PF], i32),
        "hk_test_trace_lean": ([vp, vp, i32, i32, PF, PF, PF, PF, PI, PF], i32),
        "hk_scene_bvh_info": ([vp, PI, PI, PI], i32),
        "hk_scene_light_bvh_copy": ([vp, PI, PF, C.POINTER(C.c_uint32)], i32),
    }
    for name, (args, res) in sig.items():
        fn = getattr(L, name)
        fn.argtypes = args
        fn.restype = res
    _lib = L
    return L


def check(status, what):
    if status != 0:
        msg = lib().hk_last_error()
        raise HikariMI355XError("%s failed (%d): %s" % (what, status, msg.decode() if msg else "?"))

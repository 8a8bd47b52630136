"""C-ABI misuse (-m gpu): every malformed argument is refused with a status and a message — never a crash, never a render — and the
context stays usable afterwards.  The records are corrupted one field at a time, starting from a scene that renders."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _copy(struct):
    return type(struct).from_buffer_copy(struct)


def _array_copy(ptr, n, ctype):
    arr = (ctype * n)()
    C.memmove(arr, ptr, n * C.sizeof(ctype))
    return arr


def _valid_scene(hk):
    from hikari_jl_amd import geometry as G, scenes
    s, film, cam = scenes.textured_scene(24, 24)
    fog = hk.HomogeneousMedium(sigma_a=hk.RGBSpectrum(0.1), sigma_s=hk.RGBSpectrum(0.4))
    s.push(G.rect3f((-0.2, 0.2, -0.2), (0.3, 0.3, 0.3)), hk.MediumInterface(hk.GlassMaterial(index=1.0), inside=fog))
    s.push(G.sphere((0.5, 1.2, -0.3), 0.2, 8), hk.MixMaterial((hk.MatteMaterial(), hk.Gold(roughness=0.1)), 0.3))
    s.sync()
    return s, film, cam


def test_scene_records_are_validated(hk, gpu_ctx):
    L = hk._lib.lib()
    A = hk._abi
    s, film, cam = _valid_scene(hk)
    d0 = s.desc
    out = C.c_void_p()
    assert L.hk_scene_create(gpu_ctx.h, C.byref(d0), C.byref(out)) == 0
    L.hk_scene_destroy(out)

    def refused(d, needle, keep=()):
        h = C.c_void_p()
        st = L.hk_scene_create(gpu_ctx.h, C.byref(d), C.byref(h))
        msg = L.hk_last_error() or b""
        assert st == A.HK_ERR_INVALID and needle.encode() in msg and not h.value, (needle, st, msg)

    # counts and arrays
    d = _copy(d0); d.n_materials = -1; refused(d, "negative count")
    d = _copy(d0); d.materials = None; refused(d, "null array")
    d = _copy(d0); d.positions = None; refused(d, "bad triangle arrays")
    d = _copy(d0); d.n_triangles = -5; refused(d, "bad triangle arrays")
    # triangle metadata
    meta = _array_copy(d0.meta, d0.n_triangles, A.hk_tri_meta)
    meta[3].medium_interface_idx = d0.n_media_interfaces
    d = _copy(d0); d.meta = meta; refused(d, "missing medium interface")
    meta = _array_copy(d0.meta, d0.n_triangles, A.hk_tri_meta)
    meta[0].arealight_flat_idx_1based = d0.n_lights + 1
    d = _copy(d0); d.meta = meta; refused(d, "missing area light")
    # medium interfaces
    mis = _array_copy(d0.media_interfaces, d0.n_media_interfaces, A.hk_medium_interface)
    mis[0].material = d0.n_materials
    d = _copy(d0); d.media_interfaces = mis; refused(d, "missing material")
    mis = _array_copy(d0.media_interfaces, d0.n_media_interfaces, A.hk_medium_interface)
    mis[0].inside = d0.n_media
    d = _copy(d0); d.media_interfaces = mis; refused(d, "missing medium")
    # materials: texture, spectrum and Mix child indices
    mats = _array_copy(d0.materials, d0.n_materials, A.hk_material)
    mats[0].rgb[0].tex = d0.n_textures
    d = _copy(d0); d.materials = mats; refused(d, "rgb texture index out of range")
    mats = _array_copy(d0.materials, d0.n_materials, A.hk_material)
    mats[0].f[0].tex = d0.n_textures + 7
    d = _copy(d0); d.materials = mats; refused(d, "float texture index out of range")
    mix = [i for i in range(d0.n_materials) if d0.materials[i].kind == A.HK_MAT_MIX]
    assert mix
    mats = _array_copy(d0.materials, d0.n_materials, A.hk_material)
    mats[mix[0]].i[0] = d0.n_materials
    d = _copy(d0); d.materials = mats; refused(d, "child material index out of range")
    cond = [i for i in range(d0.n_materials) if d0.materials[i].kind == A.HK_MAT_CONDUCTOR and d0.materials[i].spectrum[0] >= 0]
    if cond:
        mats = _array_copy(d0.materials, d0.n_materials, A.hk_material)
        mats[cond[0]].spectrum[0] = d0.n_spectra
        d = _copy(d0); d.materials = mats; refused(d, "spectrum index out of range")
    # textures
    tex = _array_copy(d0.textures, d0.n_textures, A.hk_texture)
    tex[0].channels = 3
    d = _copy(d0); d.textures = tex; refused(d, "bad texture record 0")
    tex = _array_copy(d0.textures, d0.n_textures, A.hk_texture)
    tex[1].width = 0
    d = _copy(d0); d.textures = tex; refused(d, "bad texture record 1")
    # lights
    lights = _array_copy(d0.lights, d0.n_lights, A.hk_light)
    lights[0].kind = 99
    d = _copy(d0); d.lights = lights; refused(d, "unknown light kind")
    # media
    media = _array_copy(d0.media, d0.n_media, A.hk_medium)
    media[0].kind = 42
    d = _copy(d0); d.media = media; refused(d, "unknown medium kind")
    # null handles / outputs
    assert L.hk_scene_create(gpu_ctx.h, None, C.byref(out)) == A.HK_ERR_INVALID
    assert L.hk_scene_create(gpu_ctx.h, C.byref(d0), None) == A.HK_ERR_INVALID
    assert L.hk_scene_create(None, C.byref(d0), C.byref(out)) == A.HK_ERR_INVALID
    # ... and after all of that the context still builds and renders the untouched scene
    vp = hk.VolPath(max_depth=3, samples=2)
    vp(s, film, cam)
    assert np.isfinite(film.framebuffer).all() and film.framebuffer.mean() > 0
    vp.close()


def test_integrator_film_and_render_arguments(hk, gpu_ctx):
    L = hk._lib.lib()
    A = hk._abi
    s, film, cam = _valid_scene(hk)
    sh = hk.scene_handle(gpu_ctx, s)

    def bad_params(**kw):
        p = hk.integrator_params()
        for k, v in kw.items():
            setattr(p, k, v)
        h = C.c_void_p()
        st = L.hk_integrator_create(gpu_ctx.h, C.byref(p), C.byref(h))
        assert st == A.HK_ERR_INVALID and not h.value, kw
        return L.hk_last_error()

    assert b"max_depth" in bad_params(max_depth=0)
    assert b"max_depth" in bad_params(max_depth=256)
    assert b"filter" in bad_params(filter_type=77)
    assert b"material_coherence" in bad_params(material_coherence=3)
    fh = C.c_void_p()
    for w, h in ((0, 8), (8, 0), (-3, 4)):
        assert L.hk_film_create(gpu_ctx.h, w, h, 0, None, C.byref(fh)) == A.HK_ERR_INVALID
    assert L.hk_film_create(gpu_ctx.h, 8, 8, 0, None, None) == A.HK_ERR_INVALID

    p = hk.integrator_params(max_depth=3, samples=4)
    integ, f32film, f64film = C.c_void_p(), C.c_void_p(), C.c_void_p()
    assert L.hk_integrator_create(gpu_ctx.h, C.byref(p), C.byref(integ)) == 0
    assert L.hk_film_create(gpu_ctx.h, 24, 24, 0, None, C.byref(f32film)) == 0
    assert L.hk_film_create(gpu_ctx.h, 24, 24, 1, None, C.byref(f64film)) == 0
    rec = cam.record()

    def render(film_h=f32film, first=1, n=1, stride=1, cam_rec=rec, scene=sh, integ_h=integ):
        return L.hk_render(gpu_ctx.h, scene, integ_h, film_h, C.byref(cam_rec) if cam_rec is not None else None, first, n, stride)

    assert render() == 0
    assert render(first=0) == A.HK_ERR_INVALID and b"sample range" in L.hk_last_error()
    assert render(n=-1) == A.HK_ERR_INVALID
    assert render(stride=0) == A.HK_ERR_INVALID
    assert render(film_h=f64film) == A.HK_ERR_INVALID and b"accumulation type" in L.hk_last_error()
    assert render(cam_rec=None) == A.HK_ERR_INVALID
    assert render(scene=None) == A.HK_ERR_INVALID
    assert render(integ_h=None) == A.HK_ERR_INVALID
    assert render(film_h=None) == A.HK_ERR_INVALID
    assert render(n=0) == 0                                                   # nothing to do is not an error
    for tile in ((-1, 0, 8, 8), (0, 0, 25, 8), (0, 0, 8, 25), (9, 0, 8, 8), (0, 9, 8, 8)):
        assert L.hk_render_tile(gpu_ctx.h, sh, integ, f32film, C.byref(rec), 1, 1, 1, *tile) == A.HK_ERR_INVALID, tile
        assert b"pixel range" in L.hk_last_error()
    assert L.hk_render_tile(gpu_ctx.h, sh, integ, f32film, C.byref(rec), 1, 1, 1, 4, 4, 4, 20) == 0   # an empty range is fine
    out = np.empty((24, 24, 3), np.float32)
    assert L.hk_film_read_rgb(gpu_ctx.h, f32film, None) == A.HK_ERR_INVALID
    assert L.hk_film_read_rgb(gpu_ctx.h, None, out.ctypes.data_as(A.PF)) == A.HK_ERR_INVALID
    # the asynchronous pair: waiting without a read in flight is an error with a message; null handles are refused; a wait may ask for neither copy nor pointer
    assert L.hk_film_read_wait(gpu_ctx.h, f64film, None, None) == A.HK_ERR_INVALID and b"hk_film_read_rgb_async" in L.hk_last_error()
    assert L.hk_film_read_rgb_async(gpu_ctx.h, None) == A.HK_ERR_INVALID and L.hk_film_read_rgb_async(None, f32film) == A.HK_ERR_INVALID
    assert L.hk_film_read_rgb(gpu_ctx.h, f32film, out.ctypes.data_as(A.PF)) == 0 and np.isfinite(out).all() and out.mean() > 0
    assert L.hk_film_read_rgb_async(gpu_ctx.h, f32film) == 0 and L.hk_film_read_wait(gpu_ctx.h, f32film, None, None) == 0
    out2, ptr = np.empty_like(out), A.PF()
    assert L.hk_film_read_rgb_async(gpu_ctx.h, f32film) == 0 and L.hk_film_read_wait(gpu_ctx.h, f32film, out2.ctypes.data_as(A.PF), C.byref(ptr)) == 0
    assert np.array_equal(out2, out) and np.array_equal(np.ctypeslib.as_array(ptr, shape=out.shape), out)
    assert L.hk_flush(None) == A.HK_ERR_INVALID and L.hk_trim_cache(None) == A.HK_ERR_INVALID and L.hk_flush(gpu_ctx.h) == 0
    assert L.hk_ctx_get_option(gpu_ctx.h, b"HK_NOT_A_KNOB", None, 0) == A.HK_ERR_INVALID
    # point-wise entry points
    z = np.zeros((4, 4), np.float32)
    PF = A.PF
    assert L.hk_test_bsdf(gpu_ctx.h, sh, 0, s.desc.n_materials, 1, 4, z.ctypes.data_as(PF), z.ctypes.data_as(PF), z.ctypes.data_as(PF), z.ctypes.data_as(PF),
                          z.ctypes.data_as(PF), z.ctypes.data_as(PF), z.ctypes.data_as(PF)) == A.HK_ERR_INVALID
    assert b"material index" in L.hk_last_error()
    assert L.hk_test_light(gpu_ctx.h, sh, 0, s.desc.n_lights + 1, 4, z.ctypes.data_as(PF), z.ctypes.data_as(PF), z.ctypes.data_as(PF), z.ctypes.data_as(PF)) == A.HK_ERR_INVALID
    assert L.hk_test_medium(gpu_ctx.h, sh, 0, s.desc.n_media, 4, z.ctypes.data_as(PF), None, None, z.ctypes.data_as(PF), z.ctypes.data_as(PF)) == A.HK_ERR_INVALID
    L.hk_film_destroy(f32film)
    L.hk_film_destroy(f64film)
    L.hk_integrator_destroy(integ)

"""VolPath integrator — host-side mirror of `Hikari.VolPath` (src/integrators/volpath/volpath.jl:29-113,
445-670) whose per-ray work runs in the HIP library behind the C-ABI (include/hikari_mi355x.h).

    vp = VolPath(max_depth=8, samples=64)      # same keywords / defaults as volpath.jl:75-84
    vp(scene, film, camera)                    # full render: reset, `samples` x render!, return film
    vp.render(scene, film, camera)             # render!: ONE more sample on top of the accumulators
    vp.clear(); vp.close()
"""
import ctypes as C

import numpy as np

from . import _abi as A
from . import _lib
from . import tables


class GaussianFilter:
    def __init__(self, radius=(1.5, 1.5), sigma=0.5):
        self.type, self.radius, self.p1, self.p2 = A.HK_FILTER_GAUSSIAN, radius, sigma, 0.0


class BoxFilter:
    def __init__(self, radius=(0.5, 0.5)):
        self.type, self.radius, self.p1, self.p2 = A.HK_FILTER_BOX, radius, 0.0, 0.0


class TriangleFilter:
    def __init__(self, radius=(2.0, 2.0)):
        self.type, self.radius, self.p1, self.p2 = A.HK_FILTER_TRIANGLE, radius, 0.0, 0.0


class MitchellFilter:
    def __init__(self, radius=(2.0, 2.0), B=1.0 / 3.0, C=1.0 / 3.0):
        self.type, self.radius, self.p1, self.p2 = A.HK_FILTER_MITCHELL, radius, B, C


class LanczosSincFilter:
    def __init__(self, radius=(4.0, 4.0), tau=3.0):
        self.type, self.radius, self.p1, self.p2 = A.HK_FILTER_LANCZOS, radius, tau, 0.0


def integrator_params(max_depth=8, samples=64, russian_roulette_depth=3, regularize=True, material_coherence="none",
                      max_component_value=10.0, filter=None, accumulation_eltype="Float32", sampler_seed=0,
                      samples_per_pass=0):
    if material_coherence not in ("none", "sorted", "per_type"):
        raise AssertionError("material_coherence must be :none, :sorted, :per_type")  # volpath.jl:85
    if accumulation_eltype not in ("Float32", "Float64"):
        raise AssertionError("accumulation_eltype must be Float32 or Float64")  # volpath.jl:86
    f = filter if filter is not None else GaussianFilter()
    p = A.hk_integrator_params()
    p.max_depth, p.samples_per_pixel, p.russian_roulette_depth = int(max_depth), int(samples), int(russian_roulette_depth)
    p.regularize = 1 if regularize else 0
    p.material_coherence = ("none", "sorted", "per_type").index(material_coherence)
    p.max_component_value = float(max_component_value)
    p.filter_type = f.type
    p.filter_radius[:] = [float(f.radius[0]), float(f.radius[1])]
    p.filter_param1, p.filter_param2 = float(f.p1), float(f.p2)
    p.accumulate_f64 = 1 if accumulation_eltype == "Float64" else 0
    p.sampler_seed = int(sampler_seed)
    p.samples_per_pass = int(samples_per_pass)
    return p


class Context:
    """One hk_ctx per GPU (tables uploaded once)."""
    _by_device = {}

    def __init__(self, device=0, stream=0):
        L = _lib.lib()
        self.h = C.c_void_p()
        _lib.check(L.hk_ctx_create(int(device), C.c_void_p(stream), C.byref(self.h)), "hk_ctx_create")
        t = tables.load()
        _lib.check(L.hk_ctx_set_tables(self.h, C.byref(t["struct"])), "hk_ctx_set_tables")
        self.device = device

    @classmethod
    def get(cls, device=0):
        if device not in cls._by_device:
            cls._by_device[device] = Context(device)
        return cls._by_device[device]

    # -- tuning knobs (hk_ctx_set_option): the library reads HK_* from the environment once, when the context is created --
    def set_option(self, name, value):
        """value None: back to the built-in default"""
        v = None if value is None else str(value).encode()
        _lib.check(_lib.lib().hk_ctx_set_option(self.h, name.encode(), v), "hk_ctx_set_option(%s)" % name)

    def get_option(self, name):
        buf = C.create_string_buffer(256)
        n = _lib.lib().hk_ctx_get_option(self.h, name.encode(), buf, 256)
        if n == A.HK_UNSET:
            return None                  # no value: the built-in default applies
        _lib.check(min(n, 0), "hk_ctx_get_option(%s)" % name)      # an unknown name raises (it used to read as "default")
        return buf.value.decode()

    def options(self, **kv):
        """with ctx.options(HK_GREY=0, HK_WAVES_PER_CU=3): ... — the knobs are restored on exit"""
        return _Options(self, kv)

    def flush(self):
        _lib.check(_lib.lib().hk_flush(self.h), "hk_flush")

    def trim_cache(self):
        _lib.check(_lib.lib().hk_trim_cache(self.h), "hk_trim_cache")


class _Options:
    def __init__(self, ctx, kv):
        self.ctx, self.kv, self.saved = ctx, kv, {}

    def __enter__(self):
        for k, v in self.kv.items():
            self.saved[k] = self.ctx.get_option(k)
            self.ctx.set_option(k, v)
        return self.ctx

    def __exit__(self, *exc):
        for k, v in self.saved.items():
            self.ctx.set_option(k, v)
        return False


def scene_handle(ctx, scene):
    d = scene.desc
    dev = getattr(scene, "_device", None)
    if dev is None:
        dev = scene._device = {}
    if id(ctx) not in dev:
        h = C.c_void_p()
        _lib.check(_lib.lib().hk_scene_create(ctx.h, C.byref(d), C.byref(h)), "hk_scene_create")
        dev[id(ctx)] = h
    return dev[id(ctx)]


class Comm:
    """hk_comm: the RCCL communicator of the film reduce (include/hikari_mi355x.h).  Comm.local([ctx, ...]) spans the GPUs of this
    process; Comm.rank(ctx, unique_id, rank, world) is one rank of a one-process-per-GPU job (unique_id from Comm.unique_id() on
    rank 0, handed to the other ranks by the launcher's side channel)."""

    def __init__(self, handle, ctxs):
        self.h, self.ctxs = handle, ctxs

    @staticmethod
    def unique_id():
        buf = (C.c_uint8 * 128)()
        _lib.check(_lib.lib().hk_comm_unique_id(buf), "hk_comm_unique_id")
        return bytes(buf)

    @classmethod
    def local(cls, ctxs):
        arr = (C.c_void_p * len(ctxs))(*[c.h for c in ctxs])
        h = C.c_void_p()
        _lib.check(_lib.lib().hk_comm_create(arr, len(ctxs), C.byref(h)), "hk_comm_create")
        return cls(h, list(ctxs))

    @classmethod
    def rank(cls, ctx, unique_id, rank, world):
        buf = (C.c_uint8 * 128).from_buffer_copy(unique_id)
        h = C.c_void_p()
        _lib.check(_lib.lib().hk_comm_create_rank(ctx.h, buf, int(rank), int(world), C.byref(h)), "hk_comm_create_rank")
        return cls(h, [ctx])

    def reduce_films(self, integrators, root=0):
        """Sum-reduce the film accumulators of `integrators` (one VolPath per local rank, in communicator order) onto rank `root`."""
        films = (C.c_void_p * len(integrators))(*[vp._film[0] for vp in integrators])
        _lib.check(_lib.lib().hk_film_reduce(self.h, films, len(integrators), int(root)), "hk_film_reduce")

    def close(self):
        if self.h:
            _lib.lib().hk_comm_destroy(self.h)
            self.h = None


class VolPath:
    def __init__(self, max_depth=8, samples=64, russian_roulette_depth=3, regularize=True, material_coherence="none",
                 max_component_value=10.0, filter=None, accumulation_eltype="Float32", device=0, samples_per_pass=0):
        self.params = integrator_params(max_depth, samples, russian_roulette_depth, regularize, material_coherence,
                                        max_component_value, filter, accumulation_eltype, 0, samples_per_pass)
        self.samples_per_pixel = int(samples)
        self.max_depth = int(max_depth)
        self.device = device
        self._ctx = None
        self._integ = None
        self._film = None      # (hk_film handle, w, h)
        self._external_accum = None
        self._readback = None  # the one host buffer hk_film_read_rgb fills (a stable pointer: read_framebuffer(view=True) pins it)
        self._pinned = None    # the film handle `_readback` is registered with (hk_film_pin_host)
        self._read_pending = False

    # -- lazily created device state (the reference's `vp.state`, volpath.jl:463-482) --
    def _ensure(self, film):
        L = _lib.lib()
        if self._ctx is None:
            self._ctx = Context.get(self.device)
        if self._integ is None:
            self._integ = C.c_void_p()
            _lib.check(L.hk_integrator_create(self._ctx.h, C.byref(self.params), C.byref(self._integ)), "hk_integrator_create")
        if self._film is None or self._film[1:] != (film.width, film.height):
            if self._film is not None:
                L.hk_film_destroy(self._film[0])       # (unregisters a pinned read-back buffer)
                self._read_pending = False
                self._pinned = None
            h = C.c_void_p()
            ext = C.c_void_p(self._external_accum) if self._external_accum else None
            _lib.check(L.hk_film_create(self._ctx.h, film.width, film.height, self.params.accumulate_f64, ext, C.byref(h)), "hk_film_create")
            self._film = (h, film.width, film.height)

    def use_external_accumulators(self, device_ptr):
        """Accumulate into caller-owned device memory (a torch tensor the host reduces with torch.distributed)."""
        self._external_accum = int(device_ptr)
        self._film = None

    def clear(self):
        if self._film is not None:
            _lib.check(_lib.lib().hk_film_clear(self._film[0]), "hk_film_clear")

    def render_samples(self, scene, film, camera, n_samples, stride=1, first=None, readback=True, tile=None):
        """Render `n_samples` more samples (sample indices first, first+stride, ...); tile = (x0, y0, x1, y1) restricts the call to
        the pixels [x0, x1) x [y0, y1) (pixel-tile sharding, hk_render_tile)."""
        self._ensure(film)
        L = _lib.lib()
        sh = scene_handle(self._ctx, scene)
        cam = camera.record()
        if first is None:
            first = film.iteration_index + 1
        if tile is None:
            _lib.check(L.hk_render(self._ctx.h, sh, self._integ, self._film[0], C.byref(cam), int(first), int(n_samples), int(stride)), "hk_render")
        else:
            _lib.check(L.hk_render_tile(self._ctx.h, sh, self._integ, self._film[0], C.byref(cam), int(first), int(n_samples), int(stride),
                                        *[int(v) for v in tile]), "hk_render_tile")
        film.iteration_index = first + (n_samples - 1) * stride
        if readback == "pipelined":
            # the frame an interactive viewer shows, one call behind: this call's samples are handed to the GPU, the frame of the PREVIOUS
            # call (its copy was enqueued then) is waited for while they render, then this call's copy is enqueued behind them
            _lib.check(L.hk_flush(self._ctx.h), "hk_flush")
            if self._read_pending:
                self._frame_from_pinned(film)
            _lib.check(L.hk_film_read_rgb_async(self._ctx.h, self._film[0]), "hk_film_read_rgb_async")
            self._read_pending = True
        elif readback:
            self.read_framebuffer(film, view=(readback == "view"))

    def render(self, scene, film, camera):
        """render!(vp, scene, film, camera): one sample, progressive (volpath.jl:445-450)."""
        self.render_samples(scene, film, camera, 1)

    def read_framebuffer(self, film, view=False):
        """film.framebuffer <- K13 of the accumulators.  The library writes Julia's [h, w] column-major matrix, i.e. C [w][h]: by default
        it is copied (transposed) into film.framebuffer; view=True makes film.framebuffer a transposed VIEW of the integrator's one
        read-back buffer instead — no host copy, and the library copies straight into that buffer once it has seen it twice
        (include/hikari_mi355x.h: hk_film_read_rgb) — valid until the next read."""
        self._read_pending = False
        rb = self._readback
        if rb is None or rb.shape != (film.width, film.height, 3):
            rb = self._readback = np.empty((film.width, film.height, 3), dtype=np.float32)
            self._pinned = None
        if view and self._pinned is not self._film[0]:
            # a viewer's loop: this integrator owns `rb` for as long as the film lives, so the library may copy straight into it
            # (hk_film_pin_host; a refusal of the driver is not an error — the read then goes through the film's staging buffers)
            if _lib.lib().hk_film_pin_host(self._film[0], rb.ctypes.data_as(A.PF)) == A.HK_OK:
                self._pinned = self._film[0]
        _lib.check(_lib.lib().hk_film_read_rgb(self._ctx.h, self._film[0], rb.ctypes.data_as(A.PF)), "hk_film_read_rgb")
        if view:
            film.framebuffer = np.transpose(rb, (1, 0, 2))
        else:
            if not film.framebuffer.flags.writeable or film.framebuffer.base is rb:
                film.framebuffer = np.empty((film.height, film.width, 3), dtype=np.float32)
            film.framebuffer[...] = np.transpose(rb, (1, 0, 2))

    def _frame_from_pinned(self, film):
        """hk_film_read_wait: film.framebuffer becomes a view of the film's pinned staging buffer holding the last asynchronous read"""
        ptr = A.PF()
        _lib.check(_lib.lib().hk_film_read_wait(self._ctx.h, self._film[0], None, C.byref(ptr)), "hk_film_read_wait")
        frame = np.ctypeslib.as_array(ptr, shape=(film.width, film.height, 3))
        film.framebuffer = np.transpose(frame, (1, 0, 2))
        self._read_pending = False

    def finish_pipelined(self, film):
        """after a loop of render_samples(..., readback="pipelined"): the frame of the LAST call"""
        if self._read_pending:
            self._frame_from_pinned(film)
            film.framebuffer = film.framebuffer.copy()      # (the pinned buffer belongs to the film)

    def read_accumulators(self, film):
        n = film.width * film.height
        dt = np.float64 if self.params.accumulate_f64 else np.float32
        out = np.empty(4 * n, dtype=dt)
        _lib.check(_lib.lib().hk_film_read_accum(self._ctx.h, self._film[0], out.ctypes.data_as(C.c_void_p)), "hk_film_read_accum")
        return out

    def __call__(self, scene, film, camera):
        film.iteration_index = 0
        self._ensure(film)
        self.clear()
        self.reset_stats()
        self.render_samples(scene, film, camera, self.samples_per_pixel)
        return film

    def stats(self):
        s = A.hk_stats()
        _lib.check(_lib.lib().hk_stats_get(self._ctx.h, C.byref(s)), "hk_stats_get")
        return s

    def reset_stats(self):
        _lib.check(_lib.lib().hk_stats_reset(self._ctx.h), "hk_stats_reset")

    def enable_counters(self, count_nodes=False, time_kernels=False):
        self._ensure_ctx()
        _lib.check(_lib.lib().hk_stats_enable_counters(self._ctx.h, (1 if count_nodes else 0) | (2 if time_kernels else 0)), "hk_stats_enable_counters")

    def _ensure_ctx(self):
        if self._ctx is None:
            self._ctx = Context.get(self.device)

    def sync(self):
        _lib.check(_lib.lib().hk_sync(self._ctx.h), "hk_sync")

    def close(self, trim_cache=False):
        """trim_cache: also give the path-state slab (kept by the library for the next integrator: up to HK_STATE_CACHE_GB) back to the driver"""
        L = _lib.lib()
        if self._film is not None:
            L.hk_film_destroy(self._film[0])
            self._film = None
        self._readback, self._read_pending, self._pinned = None, False, None
        if self._integ is not None:
            L.hk_integrator_destroy(self._integ)
            self._integ = None
        if trim_cache and self._ctx is not None:
            self._ctx.trim_cache()

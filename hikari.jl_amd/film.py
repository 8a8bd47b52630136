"""Film (src/film.jl): `framebuffer` is the linear-HDR RGB{Float32}[height, width] matrix K13 writes
(volpath.jl:415), `iteration_index` the progressive sample counter (volpath.jl:488-489)."""
import numpy as np


class Film:
    def __init__(self, resolution):
        self.resolution = (int(resolution[0]), int(resolution[1]))  # (width, height) = Point2f(w, h)
        w, h = self.resolution
        self.framebuffer = np.zeros((h, w, 3), dtype=np.float32)
        self.postprocess_buffer = np.zeros((h, w, 3), dtype=np.float32)   # film.postprocess
        self.depth = None                                                 # film.depth [h, w] when aux buffers were filled
        self.normal = None
        self.albedo = None
        self.iteration_index = 0

    @property
    def width(self):
        return self.resolution[0]

    @property
    def height(self):
        return self.resolution[1]

    def clear(self):
        self.framebuffer[...] = 0
        self.iteration_index = 0

    def fill_aux_buffers(self, scene, camera, has_infinite_lights=False, device=0):
        """fill_aux_buffers!(film, scene, camera; has_infinite_lights) (src/film.jl:410-433): film.albedo / normal / depth"""
        import ctypes as C
        from . import _abi as A
        from . import _lib
        from .volpath import Context, scene_handle
        ctx = Context.get(device)
        w, h = self.resolution
        a, n, d = np.zeros((w, h, 3), np.float32), np.zeros((w, h, 3), np.float32), np.zeros((w, h), np.float32)
        cam = camera.record()
        _lib.check(_lib.lib().hk_film_fill_aux(ctx.h, scene_handle(ctx, scene), C.byref(cam), w, h, 1 if has_infinite_lights else 0, a.ctypes.data_as(A.PF),
                                               n.ctypes.data_as(A.PF), d.ctypes.data_as(A.PF)), "hk_film_fill_aux")
        self.albedo, self.normal, self.depth = np.transpose(a, (1, 0, 2)).copy(), np.transpose(n, (1, 0, 2)).copy(), np.transpose(d, (1, 0)).copy()
        return self

    def denoise(self, config=None, device=0):
        """denoise!(film; config) (src/denoise.jl:301-376): a-trous filter of film.framebuffer guided by film.normal / film.depth
        (fill_aux_buffers first); the result goes to film.postprocess.  Like the reference, the even passes write into the
        framebuffer itself, so with iterations >= 2 film.framebuffer holds the last even pass afterwards."""
        import ctypes as C
        from . import _abi as A
        from . import _lib
        from .denoise import DenoiseConfig
        from .volpath import Context
        if getattr(self, "normal", None) is None or self.depth is None:
            raise ValueError("denoise needs film.normal and film.depth: call fill_aux_buffers first")
        p = (config or DenoiseConfig()).record()
        ctx = Context.get(device)
        h, w = self.framebuffer.shape[:2]
        src = np.ascontiguousarray(np.transpose(self.framebuffer, (1, 0, 2)), np.float32)
        nn = np.ascontiguousarray(np.transpose(self.normal, (1, 0, 2)), np.float32)
        dp = np.ascontiguousarray(np.transpose(self.depth, (1, 0)), np.float32)
        dst, after = np.empty_like(src), np.empty_like(src)
        _lib.check(_lib.lib().hk_denoise(ctx.h, C.byref(p), w, h, src.ctypes.data_as(A.PF), nn.ctypes.data_as(A.PF), dp.ctypes.data_as(A.PF),
                                         dst.ctypes.data_as(A.PF), after.ctypes.data_as(A.PF)), "hk_denoise")
        self.postprocess_buffer = np.transpose(dst, (1, 0, 2)).copy()
        self.framebuffer = np.transpose(after, (1, 0, 2)).copy()
        return self.postprocess_buffer

    def denoise_inplace(self, config=None, device=0):
        """denoise_inplace!(film; config) (src/denoise.jl:379-384): denoise!, then framebuffer <- postprocess"""
        self.denoise(config, device)
        self.framebuffer = self.postprocess_buffer.copy()
        return self.framebuffer

    def postprocess(self, exposure=1.0, tonemap="aces", gamma=2.2, white_point=4.0, sensor=None, background=None, device=0):
        """postprocess!(film; exposure, tonemap, gamma, white_point, sensor, background) (src/postprocess.jl:293-357): reads
        film.framebuffer, writes and returns film.postprocess; non-destructive, runs on the device through the C-ABI."""
        import ctypes as C
        from . import _abi as A
        from . import _lib
        from .postprocess import make_params
        from .volpath import Context
        p = make_params(exposure, tonemap, gamma, white_point, sensor, background)
        ctx = Context.get(device)
        h, w = self.framebuffer.shape[:2]
        src = np.ascontiguousarray(np.transpose(self.framebuffer, (1, 0, 2)), np.float32)
        dst = np.empty_like(src)
        dp = None
        if background is not None and self.depth is not None:
            dp = np.ascontiguousarray(np.transpose(self.depth, (1, 0)), np.float32)
        _lib.check(_lib.lib().hk_postprocess(ctx.h, C.byref(p), w, h, src.ctypes.data_as(A.PF), dp.ctypes.data_as(A.PF) if dp is not None else None,
                                             dst.ctypes.data_as(A.PF)), "hk_postprocess")
        self.postprocess_buffer = np.transpose(dst, (1, 0, 2)).copy()
        return self.postprocess_buffer

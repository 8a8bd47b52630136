"""Film (src/film.jl): `framebuffer` is the linear-HDR RGB{Float32}[height, width] matrix K13 writes
(volpath.jl:415), `iteration_index` the progressive sample counter (volpath.jl:488-489)."""
import numpy as np


class Film:
    def __init__(self, resolution):
        self.resolution = (int(resolution[0]), int(resolution[1]))  # (width, height) = Point2f(w, h)
        w, h = self.resolution
        self.framebuffer = np.zeros((h, w, 3), dtype=np.float32)
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

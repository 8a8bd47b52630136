"""PerspectiveCamera (src/camera/perspective.jl:1-128): builds raster_to_camera / camera_to_world in float32
the way ProjectiveCamera does and exposes them as one hk_camera record (the C-ABI carries matrices only)."""
import numpy as np

from . import _abi as A
from . import geometry as G

f32 = np.float32


class PerspectiveCamera:
    def __init__(self, eyepos, lookat, film, up=(0, 1, 0), fov=55.0, screen_window=((-1, -1), (1, 1)),
                 lens_radius=0.0, focal_distance=1e6, shutter_open=0.0, shutter_close=1.0):
        # PerspectiveCamera(eyepos, lookat, film; up, fov): screen = Bounds2(-1, 1) (perspective.jl:88-94)
        world_to_camera = G.look_at(eyepos, lookat, up)
        self.camera_to_world = G.inv(world_to_camera)
        camera_to_screen = G.perspective(fov, 0.01, 1000.0)
        (x0, y0), (x1, y1) = screen_window
        w, h = film.resolution
        resolution = G.scale(w, h, 1)
        inv_bounds = G.scale(f32(1) / f32(x1 - x0), f32(1) / f32(y1 - y0), 1)
        offset = G.translate((-x0, -y0, 0))
        raster_to_screen = (G.inv(offset) @ G.inv(inv_bounds) @ G.inv(resolution)).astype(f32)
        self.raster_to_camera = (G.inv(camera_to_screen) @ raster_to_screen).astype(f32)
        self.lens_radius, self.focal_distance = float(lens_radius), float(focal_distance)
        self.shutter_open, self.shutter_close = float(shutter_open), float(shutter_close)
        p_min = self._apply_point(self.raster_to_camera, (0, 0, 0))
        self.dx_camera = self._apply_point(self.raster_to_camera, (1, 0, 0)) - p_min
        self.dy_camera = self._apply_point(self.raster_to_camera, (0, 1, 0)) - p_min
        self.film = film

    @staticmethod
    def _apply_point(m, p):
        v = m @ np.array([p[0], p[1], p[2], 1], dtype=f32)
        return (v[:3] / v[3]).astype(f32) if v[3] != 1 else v[:3].astype(f32)

    def record(self):
        c = A.hk_camera()
        c.raster_to_camera[:] = [float(x) for x in self.raster_to_camera.reshape(-1)]
        c.camera_to_world[:] = [float(x) for x in self.camera_to_world.reshape(-1)]
        c.lens_radius, c.focal_distance = self.lens_radius, self.focal_distance
        c.shutter_open, c.shutter_close = self.shutter_open, self.shutter_close
        c.dx_camera[:] = [float(x) for x in self.dx_camera]
        c.dy_camera[:] = [float(x) for x in self.dy_camera]
        return c

"""PerspectiveCamera (src/camera/perspective.jl:1-128) and MatrixCamera (src/camera/matrix.jl:13-115): builds raster_to_camera / camera_to_world in float32
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
        # Raycore's look_at / perspective are not in the reference tree.  generate_ray's lens branch states their convention
        # (perspective.jl:109: "Camera looks in -z, so d[3] < 0; negate to get positive t"), so the mirror builds that frame: pbrt's
        # LookAt / Perspective with the camera's z axis reversed.  Negating the z column of camera_to_world and the z row of
        # raster_to_camera leaves every pinhole ray bit-identical (two exact sign flips cancel); with a lens, t = -focal / d.z is
        # now positive and the ray passes through the plane of focus in FRONT of the camera instead of behind it.
        flip = np.diag(np.array([1, 1, -1, 1], dtype=f32))
        self.camera_to_world = (self.camera_to_world @ flip).astype(f32)
        self.raster_to_camera = (flip @ self.raster_to_camera).astype(f32)
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


class MatrixCamera(PerspectiveCamera):
    """MatrixCamera(view, projection, resolution[, screen_window]) (src/camera/matrix.jl:30-97): Makie-style world-to-camera
    `view` and camera-to-clip `projection` (OpenGL conventions, camera looks along -z).  camera_to_world = inv(view),
    raster_to_camera = inv(projection) * raster_to_screen; no depth of field (generate_ray, matrix.jl:99-115, is the
    PerspectiveCamera ray without a lens), shutter 0..1 — so it is the same hk_camera record with lens_radius = 0."""

    def __init__(self, view, projection, resolution, screen_window=((-1, -1), (1, 1))):
        view = np.asarray(view, dtype=np.float64).astype(f32)            # Mat4f(view): Makie stores Float64
        projection = np.asarray(projection, dtype=np.float64).astype(f32)
        self.camera_to_world = G.inv(view)
        (x0, y0), (x1, y1) = screen_window
        w, h = (resolution.resolution if hasattr(resolution, "resolution") else resolution)
        res_scale = G.scale(w, h, 1)
        inv_bounds = G.scale(f32(1) / f32(x1 - x0), f32(1) / f32(y1 - y0), 1)
        offset = G.translate((-x0, -y0, 0))
        raster_to_screen = (G.inv(offset) @ G.inv(inv_bounds) @ G.inv(res_scale)).astype(f32)
        self.raster_to_camera = (G.inv(projection) @ raster_to_screen).astype(f32)
        self.lens_radius, self.focal_distance = 0.0, 1e6
        self.shutter_open, self.shutter_close = 0.0, 1.0
        p_min = self._apply_point(self.raster_to_camera, (0, 0, 0))
        p_max = self._apply_point(self.raster_to_camera, (w, h, 0))
        self.dx_camera = self._apply_point(self.raster_to_camera, (1, 0, 0)) - p_min
        self.dy_camera = self._apply_point(self.raster_to_camera, (0, 1, 0)) - p_min
        pp = p_min[:2] / p_min[2] - p_max[:2] / p_max[2]
        self.A = float(abs(pp[0] * pp[1]))
        self.film = resolution if hasattr(resolution, "resolution") else None

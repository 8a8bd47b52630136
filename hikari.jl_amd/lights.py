"""Host-side mirror of Hikari's light records (src/lights/point.jl, spot.jl, directional.jl, sun.jl, ambient.jl).
The reference picks `scale` by *which constructor* is used (quirk Q4): the (position, ::Spectrum) forms keep
scale = 1; the `RGBSpectrum-first` forms use 1/D65_PHOTOMETRIC; the `RGB` forms bake an RGBIlluminantSpectrum
and use 1/10567 too.  Python has no dispatch on argument order, so each variant is a named constructor."""
import numpy as np

from . import _abi as A
from . import geometry as G
from .materials import RGBSpectrum
from .tables import rgb_to_spectrum

f32 = np.float32
D65_PHOTOMETRIC = 10567.0


class RGBIlluminantSpectrum:
    """rgb_illuminant_spectrum(table, r, g, b) (spectral/rgb2spec.jl:371-385)."""

    def __init__(self, r, g, b):
        m = max(float(r), float(g), float(b))
        if m <= 0:
            self.poly, self.scale = (0.0, 0.0, -1e10), 0.0
        else:
            s = f32(2) * f32(m)
            self.poly = tuple(float(v) for v in rgb_to_spectrum(f32(r) / s, f32(g) / s, f32(b) / s))
            self.scale = float(s)


class Light:
    kind = -1


def _spec_fields(i):
    if isinstance(i, RGBIlluminantSpectrum):
        return dict(spectrum_kind=A.HK_SPEC_ILLUMINANT, poly=i.poly, illum_scale=i.scale, i_rgb=(0, 0, 0, 1))
    return dict(spectrum_kind=A.HK_SPEC_RGB, poly=(0, 0, 0), illum_scale=0.0, i_rgb=i.c)


class PointLight(Light):
    kind = A.HK_LIGHT_POINT

    def __init__(self, position, i, scale=1.0):
        """PointLight(position, i::Spectrum, scale=1f0)  (point.jl:26-28)"""
        self.position, self.i, self.scale = tuple(float(f32(v)) for v in position), i, float(f32(scale))

    @classmethod
    def from_spectrum_first(cls, i, position):
        """PointLight(i::RGBSpectrum, position): scale = 1/D65_PHOTOMETRIC (point.jl:68-71)"""
        return cls(position, i, f32(1) / f32(D65_PHOTOMETRIC))

    @classmethod
    def from_rgb(cls, rgb, position, power=None):
        """PointLight(rgb::RGB, position; power) (point.jl:55-66)"""
        sp = RGBIlluminantSpectrum(*rgb)
        scale = f32(1) / f32(D65_PHOTOMETRIC)
        if power is not None:
            scale = scale * f32(power) / (f32(4) * f32(np.pi))
        return cls(position, sp, scale)


class SpotLight(Light):
    kind = A.HK_LIGHT_SPOT

    def __init__(self, position, target, i, total_width, falloff_start, scale=1.0):
        """SpotLight(position, target, i, total_width, falloff_start, scale=1f0) (spot.jl)"""
        self.position = tuple(float(f32(v)) for v in position)
        # _spotlight_transform: light space looks down +z towards the target
        w2l = G.look_at(position, target, (0, 1, 0) if abs(G.normalize(np.subtract(target, position))[1]) < 0.999 else (1, 0, 0))
        self.world_to_light = w2l
        self.light_to_world = G.inv(w2l)
        self.i, self.scale = i, float(f32(scale))
        self.cos_total_width = float(np.cos(np.deg2rad(f32(total_width)), dtype=f32))
        self.cos_falloff_start = float(np.cos(np.deg2rad(f32(falloff_start)), dtype=f32))


class DirectionalLight(Light):
    kind = A.HK_LIGHT_DIRECTIONAL

    def __init__(self, i, direction, scale=1.0):
        """DirectionalLight(Transformation(), i::Spectrum, direction, scale=1f0) (directional.jl:19-27)"""
        self.i, self.scale = i, float(f32(scale))
        self.direction = tuple(float(v) for v in G.normalize(direction))

    @classmethod
    def from_spectrum_first(cls, i, direction):
        return cls(i, direction, f32(1) / f32(D65_PHOTOMETRIC))

    @classmethod
    def from_rgb(cls, rgb, direction, illuminance=None):
        scale = f32(1) / f32(D65_PHOTOMETRIC)
        if illuminance is not None:
            scale = scale * f32(illuminance)
        return cls(RGBIlluminantSpectrum(*rgb), direction, scale)


class SunLight(DirectionalLight):
    kind = A.HK_LIGHT_SUN


class AmbientLight(Light):
    kind = A.HK_LIGHT_AMBIENT

    def __init__(self, i, scale=1.0):
        """AmbientLight(s::Spectrum) = AmbientLight(s, 1f0) (ambient.jl)"""
        self.i, self.scale = i, float(f32(scale))

    @classmethod
    def from_rgb(cls, rgb):
        return cls(RGBIlluminantSpectrum(*rgb), f32(1) / f32(D65_PHOTOMETRIC))


class DiffuseAreaLight(Light):
    """Per-emissive-triangle light registered by Scene.push (scene-mesh.jl:98-131)."""
    kind = A.HK_LIGHT_DIFFUSE_AREA

    def __init__(self, vertices, normal, area, uv, Le, scale, two_sided):
        self.vertices, self.normal, self.area, self.uv = vertices, normal, area, uv
        self.Le, self.scale, self.two_sided = Le, scale, two_sided

"""Host-side mirror of Hikari's material / texture parameter containers (src/materials/uber-material.jl:180-215,
378-384, coated-diffuse.jl, coated-conductor.jl:48-77, thin-dielectric.jl:45-47, diffuse-transmission.jl:39-43,
coated-diffuse-transmission.jl:12-24, mix-material.jl, emissive.jl:43-60, medium-interface.jl:39-56).
These are plain parameter records: same names, same keyword defaults; all BSDF work happens on the device."""
import numpy as np

from . import _abi as A

f32 = np.float32


class RGBSpectrum:
    """r, g, b, alpha (src/spectrum.jl:38-43)."""

    def __init__(self, r=0.0, g=None, b=None, a=1.0):
        if g is None:
            g = b = r
        self.c = (float(f32(r)), float(f32(g)), float(f32(b)), float(f32(a)))

    def __mul__(self, s):
        return RGBSpectrum(*(float(f32(x) * f32(s)) for x in self.c))

    __rmul__ = __mul__

    def __repr__(self):
        return "RGBSpectrum%s" % (self.c,)


class Texture:
    """2-D texture (textures/basic.jl): data[h, w] floats or data[h, w, 4] RGBA, sampled bilinearly."""

    def __init__(self, data):
        d = np.asarray(data, dtype=f32)
        if d.ndim == 3 and d.shape[2] == 3:
            d = np.concatenate([d, np.ones(d.shape[:2] + (1,), dtype=f32)], axis=2)
        self.data = d  # [h, w] or [h, w, 4]


class VertexColorTexture(Texture):
    """VertexColorTexture(face_colors[3, n_faces]) (textures/basic.jl:42-46): three RGB(A) colours per face, interpolated
    with the hit's barycentrics (texture-ref.jl:230-235).  `face_colors` here is [n_faces, 3, 3|4]."""

    def __init__(self, face_colors):
        d = np.asarray(face_colors, dtype=f32)
        assert d.ndim == 3 and d.shape[1] == 3 and d.shape[2] in (3, 4)
        if d.shape[2] == 3:
            d = np.concatenate([d, np.ones(d.shape[:2] + (1,), dtype=f32)], axis=2)
        self.data = np.ascontiguousarray(d)
        self.n_faces = d.shape[0]


def _rgb(v):
    if isinstance(v, (RGBSpectrum, Texture)):
        return v
    if np.isscalar(v):
        return RGBSpectrum(v)
    return RGBSpectrum(*v)


class Material:
    kind = A.HK_MAT_FALLBACK


class MatteMaterial(Material):
    kind = A.HK_MAT_MATTE

    def __init__(self, Kd=RGBSpectrum(0.5), sigma=0.0):
        self.Kd, self.sigma = _rgb(Kd), sigma


class MirrorMaterial(Material):
    kind = A.HK_MAT_MIRROR

    def __init__(self, Kr=RGBSpectrum(0.9)):
        self.Kr = _rgb(Kr)


class GlassMaterial(Material):
    kind = A.HK_MAT_GLASS

    def __init__(self, Kr=RGBSpectrum(1.0), Kt=RGBSpectrum(1.0), u_roughness=0.0, v_roughness=0.0, index=1.5,
                 remap_roughness=True):
        self.Kr, self.Kt, self.index = _rgb(Kr), _rgb(Kt), index
        self.u_roughness, self.v_roughness, self.remap_roughness = u_roughness, v_roughness, remap_roughness


class PiecewiseLinearSpectrum:
    def __init__(self, lambdas, values):
        self.lambdas = np.ascontiguousarray(lambdas, dtype=f32)
        self.values = np.ascontiguousarray(values, dtype=f32)


class ConductorMaterial(Material):
    kind = A.HK_MAT_CONDUCTOR

    def __init__(self, eta=RGBSpectrum(0.2, 0.2, 0.2), k=RGBSpectrum(3.9, 3.9, 3.9), roughness=0.1,
                 reflectance=RGBSpectrum(1.0), remap_roughness=True):
        """keyword constructor defaults of uber-material.jl:418-426"""
        self.eta = eta if isinstance(eta, PiecewiseLinearSpectrum) else _rgb(eta)
        self.k = k if isinstance(k, PiecewiseLinearSpectrum) else _rgb(k)
        self.roughness, self.reflectance, self.remap_roughness = roughness, _rgb(reflectance), remap_roughness


def _metal_spectra():
    """measured eta/k of Ag, Al, Au, Cu, CuZn (spectral/metal-spectra.jl; data/metal_spectra.bin)"""
    import os
    import struct
    if not _METALS:
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "metal_spectra.bin")
        with open(path, "rb") as f:
            (count,) = struct.unpack("<i", f.read(4))
            for _ in range(count):
                (ln,) = struct.unpack("<i", f.read(4))
                name = f.read(ln).decode()
                (n,) = struct.unpack("<i", f.read(4))
                lam = np.frombuffer(f.read(4 * n), dtype=f32).copy()
                val = np.frombuffer(f.read(4 * n), dtype=f32).copy()
                _METALS[name] = PiecewiseLinearSpectrum(lam, val)
    return _METALS


_METALS = {}


def _metal(prefix, roughness, reflectance, remap_roughness):
    m = _metal_spectra()
    return ConductorMaterial(m[prefix + "_ETA_SPECTRUM"], m[prefix + "_K_SPECTRUM"], roughness, reflectance, remap_roughness)


def Gold(roughness=0.0, reflectance=RGBSpectrum(1.0), remap_roughness=True):
    """Gold(; roughness=0, reflectance=(1,1,1), remap_roughness=true) (uber-material.jl:469-470): measured Au eta/k"""
    return _metal("AU", roughness, reflectance, remap_roughness)


def Silver(roughness=0.0, reflectance=RGBSpectrum(1.0), remap_roughness=True):
    return _metal("AG", roughness, reflectance, remap_roughness)


def Copper(roughness=0.0, reflectance=RGBSpectrum(1.0), remap_roughness=True):
    return _metal("CU", roughness, reflectance, remap_roughness)


def Aluminum(roughness=0.0, reflectance=RGBSpectrum(1.0), remap_roughness=True):
    return _metal("AL", roughness, reflectance, remap_roughness)


def Brass(roughness=0.0, reflectance=RGBSpectrum(1.0), remap_roughness=True):
    return _metal("CUZN", roughness, reflectance, remap_roughness)


class CoatedDiffuseMaterial(Material):
    kind = A.HK_MAT_COATED_DIFFUSE

    def __init__(self, reflectance=RGBSpectrum(0.5), u_roughness=0.0, v_roughness=0.0, thickness=0.01, eta=1.5,
                 albedo=RGBSpectrum(0.0), g=0.0, max_depth=10, n_samples=1, remap_roughness=True):
        self.reflectance, self.albedo = _rgb(reflectance), _rgb(albedo)
        self.u_roughness, self.v_roughness, self.thickness, self.eta, self.g = u_roughness, v_roughness, thickness, eta, g
        self.max_depth, self.n_samples, self.remap_roughness = max_depth, n_samples, remap_roughness


PlasticMaterial = CoatedDiffuseMaterial


class ThinDielectricMaterial(Material):
    kind = A.HK_MAT_THIN_DIELECTRIC

    def __init__(self, eta=1.5):
        self.eta = eta


class DiffuseTransmissionMaterial(Material):
    kind = A.HK_MAT_DIFFUSE_TRANSMISSION

    def __init__(self, reflectance=RGBSpectrum(0.25), transmittance=RGBSpectrum(0.25), scale=1.0):
        self.reflectance, self.transmittance, self.scale = _rgb(reflectance), _rgb(transmittance), scale


class CoatedDiffuseTransmissionMaterial(Material):
    kind = A.HK_MAT_COATED_DIFFUSE_TRANSMISSION

    def __init__(self, reflectance=RGBSpectrum(0.25), transmittance=RGBSpectrum(0.25), u_roughness=0.0,
                 v_roughness=0.0, thickness=0.01, eta=1.5, albedo=RGBSpectrum(0.0), g=0.0, max_depth=10, n_samples=1,
                 remap_roughness=True):
        self.reflectance, self.transmittance, self.albedo = _rgb(reflectance), _rgb(transmittance), _rgb(albedo)
        self.u_roughness, self.v_roughness, self.thickness, self.eta, self.g = u_roughness, v_roughness, thickness, eta, g
        self.max_depth, self.n_samples, self.remap_roughness = max_depth, n_samples, remap_roughness


class CoatedConductorMaterial(Material):
    kind = A.HK_MAT_COATED_CONDUCTOR

    def __init__(self, interface_u_roughness=0.0, interface_v_roughness=0.0, interface_eta=1.5,
                 conductor_eta=None, conductor_k=None, reflectance=None, conductor_u_roughness=0.0,
                 conductor_v_roughness=0.0, thickness=0.01, albedo=RGBSpectrum(0.0), g=0.0, max_depth=10, n_samples=1,
                 remap_roughness=True):
        self.interface_u_roughness, self.interface_v_roughness, self.interface_eta = interface_u_roughness, interface_v_roughness, interface_eta
        self.use_eta_k = reflectance is None
        self.conductor_eta = conductor_eta if conductor_eta is not None else RGBSpectrum(0.2, 0.92, 1.1)
        self.conductor_k = conductor_k if conductor_k is not None else RGBSpectrum(3.9, 2.45, 2.14)
        self.reflectance = _rgb(reflectance) if reflectance is not None else RGBSpectrum(1.0)
        self.conductor_u_roughness, self.conductor_v_roughness = conductor_u_roughness, conductor_v_roughness
        self.thickness, self.albedo, self.g = thickness, _rgb(albedo), g
        self.max_depth, self.n_samples, self.remap_roughness = max_depth, n_samples, remap_roughness


class Emissive(Material):
    """Emission data (emissive.jl:43-60).  As a *surface material* it has no BSDF method, so the path
    falls back to the gray 0.5 Lambertian (quirk Q24); its triangles register DiffuseAreaLights."""
    kind = A.HK_MAT_FALLBACK

    def __init__(self, Le=RGBSpectrum(1.0), scale=1.0, two_sided=False):
        self.Le, self.scale, self.two_sided = _rgb(Le), float(scale), bool(two_sided)


class MixMaterial(Material):
    kind = A.HK_MAT_MIX

    def __init__(self, materials, amount=0.5):
        self.material1, self.material2 = materials
        self.amount = amount


class MediumInterface(Material):
    """Material wrapper: BSDF material + inside/outside media + optional emission (medium-interface.jl:39-56)."""

    def __init__(self, material, inside=None, outside=None, emission=None):
        self.material, self.inside, self.outside, self.emission = material, inside, outside, emission

"""Named homogeneous media (src/integrators/volpath/media.jl:1758-2030): the measured sigma_s / sigma_a table of pbrt-v4's named
media (Jensen et al. 2001; Narasimhan et al. 2006, mm^-1) plus the convenience constructors built on it.  The numbers are the
reference's table, extracted by tools/extract_reference_tables.py into data/medium_presets.json (SHA pinned in
tests/golden/data_tables.json)."""
import json
import os

import numpy as np

from .materials import RGBSpectrum
from .media import HomogeneousMedium

f32 = np.float32
_PRESETS = None


def _presets():
    global _PRESETS
    if _PRESETS is None:
        with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "medium_presets.json")) as f:
            _PRESETS = json.load(f)
    return _PRESETS


def get_medium_preset(name):
    """get_medium_preset(name) -> {"sigma_s": (r, g, b), "sigma_a": (r, g, b)}   (media.jl:1851-1854)"""
    p = _presets()
    if name not in p:
        raise KeyError("Unknown medium preset: %s. Available: %s" % (name, sorted(p)))
    return p[name]


def _from_preset(name, scale, g):
    props = get_medium_preset(name)
    return HomogeneousMedium(sigma_a=RGBSpectrum(*props["sigma_a"]) * f32(scale), sigma_s=RGBSpectrum(*props["sigma_s"]) * f32(scale), g=float(f32(g)))


def Milk(scale=1.0, g=0.0):
    """Milk(; scale, g)  (media.jl:1874-1879)"""
    return _from_preset("Wholemilk", scale, g)


def Smoke(density=0.5, albedo=0.9, g=0.0):
    """Smoke(; density, albedo, g): gray, sigma_t = density, sigma_s = sigma_t * albedo  (media.jl:1900-1908)"""
    st = f32(density)
    return HomogeneousMedium(sigma_a=RGBSpectrum(float(st * (f32(1) - f32(albedo)))), sigma_s=RGBSpectrum(float(st * f32(albedo))), g=float(f32(g)))


def Fog(density=0.1, g=0.0):
    """Fog(; density, g): sigma_s = density, sigma_a = density * 0.001  (media.jl:1926-1931)"""
    return HomogeneousMedium(sigma_a=RGBSpectrum(float(f32(density) * f32(0.001))), sigma_s=RGBSpectrum(float(f32(density))), g=float(f32(g)))


_JUICES = {"apple": "AppleJuice", "cranberry": "CranberryJuice", "grape": "GrapeJuice", "grapefruit": "RubyGrapefruitJuice"}
_WINES = {"chardonnay": "Chardonnay", "zinfandel": "WhiteZinfandel", "merlot": "Merlot"}


def Juice(name, scale=1.0, g=0.0):
    """Juice(name::Symbol; scale, g)  (media.jl:1946-1962)"""
    if name not in _JUICES:
        raise KeyError("Unknown juice type: %s. Available: %s" % (name, sorted(_JUICES)))
    return _from_preset(_JUICES[name], scale, g)


def Wine(name, scale=1.0, g=0.0):
    """Wine(name::Symbol; scale, g)  (media.jl:1978-1992)"""
    if name not in _WINES:
        raise KeyError("Unknown wine type: %s. Available: %s" % (name, sorted(_WINES)))
    return _from_preset(_WINES[name], scale, g)


def Coffee(scale=1.0, g=0.0):
    """Coffee(; scale, g)  (media.jl:2006-2011)"""
    return _from_preset("Espresso", scale, g)


def SubsurfaceMedium(name, scale=1.0, g=0.0):
    """SubsurfaceMedium(name::String; scale, g)  (media.jl:2027-2032)"""
    return _from_preset(name, scale, g)

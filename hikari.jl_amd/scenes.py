"""Scene scripts for BASELINE.json's configs, written against the host-side mirror of Hikari's API the
way the reference's own examples/tests are written (examples/single_triangle_test.jl:12-88,
test/volpath_integration.jl:9-115; sizes and parameters per SURVEY.md §8d)."""
import numpy as np

from . import geometry as G
from .camera import PerspectiveCamera
from .film import Film
from .lights import DirectionalLight, PointLight
from .materials import (ConductorMaterial, Emissive, GlassMaterial, MatteMaterial, MediumInterface, MirrorMaterial,
                        RGBSpectrum)
from .scene import Scene


def single_triangle(width=800, height=600):
    """Config 1 (plumbing): examples/single_triangle_test.jl:12-88 restated in the current API."""
    from .geometry import Mesh
    s = Scene()
    mesh = Mesh([[(-1, -0.5, 0), (1, -0.5, 0), (0, 1, 0)]], [[(0, 0, 1), (0.7, 0, 0.714), (0, 0.7, 0.714)]],
                [[(0, 0), (1, 0), (0.5, 1)]])
    s.push(mesh, MatteMaterial(Kd=RGBSpectrum(0.8)))
    s.push(DirectionalLight(RGBSpectrum(2.0), (0, 0, -1)))  # 3-arg inner ctor => scale = 1
    s.sync()
    film = Film((width, height))
    aspect = width / height
    cam = PerspectiveCamera((0, 0, 3), (0, 0, 0), film, up=(0, 1, 0), fov=50.0,
                            screen_window=((-aspect, -1), (aspect, 1)))
    return s, film, cam


def cornell_box(width=800, height=800, light="area", spheres=True, tess=32):
    """Config 2: Cornell box, diffuse + area light: box 2x2x2 from 0.01-thick slabs, white .73,
    left red (.65,.05,.05), right green (.12,.45,.15), two matte objects, 0.5x0.5 quad light at y=1.98 facing -y
    with Emissive(Le=1, scale=1, two_sided=false); camera (0,1,-3.5)->(0,1,0), fov 40."""
    white = MatteMaterial(Kd=RGBSpectrum(0.73, 0.73, 0.73))
    red = MatteMaterial(Kd=RGBSpectrum(0.65, 0.05, 0.05))
    green = MatteMaterial(Kd=RGBSpectrum(0.12, 0.45, 0.15))
    box, half = 2.0, 1.0
    s = Scene()
    if light == "point":
        s.push(PointLight((0, 1.8, 0), RGBSpectrum(15.0)))
    s.push(G.rect3f((-half, 0, -half), (box, 0.01, box)), white)             # floor
    s.push(G.rect3f((-half, box - 0.01, -half), (box, 0.01, box)), white)    # ceiling
    s.push(G.rect3f((-half, 0, half - 0.01), (box, box, 0.01)), white)       # back
    s.push(G.rect3f((-half, 0, -half), (0.01, box, box)), red)               # left
    s.push(G.rect3f((half - 0.01, 0, -half), (0.01, box, box)), green)       # right
    if spheres:
        s.push(G.sphere((-0.4, 0.4, 0.0), 0.35, tess), white)
        s.push(G.rect3f((0.15, 0.0, -0.1), (0.5, 0.6, 0.5)), white)
    if light == "area":
        y = 1.98
        q = G.quad((-0.25, y, -0.25), (0.25, y, -0.25), (0.25, y, 0.25), (-0.25, y, 0.25), normal=(0, -1, 0))
        s.push(q, MediumInterface(MatteMaterial(Kd=RGBSpectrum(0.0)), emission=Emissive(Le=RGBSpectrum(1.0), scale=1.0, two_sided=False)))
    s.sync()
    film = Film((width, height))
    cam = PerspectiveCamera((0, 1, -3.5), (0, 1, 0), film, fov=40.0)
    return s, film, cam


def integration_test_scene(width=64, height=64, with_fog=False):
    """test/volpath_integration.jl:9-115 (fog medium optional until the media rows land)."""
    white = MatteMaterial(Kd=RGBSpectrum(0.73, 0.73, 0.73))
    red = MatteMaterial(Kd=RGBSpectrum(0.65, 0.05, 0.05))
    green = MatteMaterial(Kd=RGBSpectrum(0.12, 0.45, 0.15))
    glass = GlassMaterial(Kr=RGBSpectrum(1.0), Kt=RGBSpectrum(1.0), index=1.5)
    gold = ConductorMaterial(eta=RGBSpectrum(0.15557, 0.42415, 1.3831), k=RGBSpectrum(3.6024, 2.4721, 1.9155))
    box, half = 2.0, 1.0
    s = Scene()
    s.push(G.rect3f((-half, 0, -half), (box, 0.01, box)), white)
    s.push(G.rect3f((-half, 0, half - 0.01), (box, box, 0.01)), white)
    s.push(G.rect3f((-half, 0, -half), (0.01, box, box)), red)
    s.push(G.rect3f((half - 0.01, 0, -half), (0.01, box, box)), green)
    s.push(G.sphere((-0.4, 0.4, 0.0), 0.35, 32), glass)
    s.push(G.sphere((0.4, 0.35, 0.0), 0.3, 32), gold)
    s.push(PointLight((0, 1.8, 0), RGBSpectrum(15.0)))
    s.sync()
    film = Film((width, height))
    cam = PerspectiveCamera((0, 1, -3.5), (0, 1, 0), film, fov=40.0)
    return s, film, cam

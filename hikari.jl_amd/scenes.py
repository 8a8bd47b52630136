"""Scene scripts for BASELINE.json's configs, written against the host-side mirror of Hikari's API the
way the reference's own examples/tests are written (examples/single_triangle_test.jl:12-88,
test/volpath_integration.jl:9-115; sizes and parameters per SURVEY.md §8d)."""
import numpy as np

f32 = np.float32

from . import geometry as G
from .camera import PerspectiveCamera
from .film import Film
from .lights import AmbientLight, DirectionalLight, PointLight, SpotLight
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


def cornell_box(width=800, height=800, light="area", spheres=True, tess=32, objects="sphere_box", object_material=None):
    """Config 2: Cornell box, diffuse + area light: box 2x2x2 from 0.01-thick slabs, white .73,
    left red (.65,.05,.05), right green (.12,.45,.15), two matte objects, 0.5x0.5 quad light at y=1.98 facing -y
    with Emissive(Le=1, scale=1, two_sided=false); camera (0,1,-3.5)->(0,1,0), fov 40.
    objects = "sphere_box" (rounds 1-3: one sphere tessellated at `tess` + one box, 1 934 triangles) or "two_spheres" (the two
    spheres of test/volpath_integration.jl:58-62, both tessellated at `tess`: SURVEY 8(d)'s "~4 k triangles").  object_material: the
    material of the first sphere instead of white matte (the per-pixel pins put a rough conductor there)."""
    white = MatteMaterial(Kd=RGBSpectrum(0.73, 0.73, 0.73))
    red = MatteMaterial(Kd=RGBSpectrum(0.65, 0.05, 0.05))
    green = MatteMaterial(Kd=RGBSpectrum(0.12, 0.45, 0.15))
    box, half = 2.0, 1.0
    s = Scene()
    if light in ("ambient", "all"):
        s.push(AmbientLight(RGBSpectrum(0.5, 0.6, 0.9)))
    if light in ("spot", "all"):     # from the front top left onto the objects, 25 degree cone with a soft edge from 15
        s.push(SpotLight((-0.7, 1.7, -0.8), (0.1, 0.3, 0.1), RGBSpectrum(20.0, 18.0, 14.0), 25.0, 15.0))
    if light in ("dir", "all"):      # into the open front of the box, downwards
        s.push(DirectionalLight(RGBSpectrum(2.0, 1.9, 1.6), (0.25, -0.45, 1.0)))
    if light in ("point", "both", "all"):
        s.push(PointLight((0, 1.8, 0) if light == "point" else (0.5, 1.6, -0.4), RGBSpectrum(15.0) if light == "point" else RGBSpectrum(6.0, 5.0, 3.0)))
    s.push(G.rect3f((-half, 0, -half), (box, 0.01, box)), white)             # floor
    s.push(G.rect3f((-half, box - 0.01, -half), (box, 0.01, box)), white)    # ceiling
    s.push(G.rect3f((-half, 0, half - 0.01), (box, box, 0.01)), white)       # back
    s.push(G.rect3f((-half, 0, -half), (0.01, box, box)), red)               # left
    s.push(G.rect3f((half - 0.01, 0, -half), (0.01, box, box)), green)       # right
    if spheres:
        s.push(G.sphere((-0.4, 0.4, 0.0), 0.35, tess), object_material if object_material is not None else white)
        if objects == "two_spheres":
            s.push(G.sphere((0.4, 0.35, 0.0), 0.3, tess), white)
        else:
            s.push(G.rect3f((0.15, 0.0, -0.1), (0.5, 0.6, 0.5)), white)
    if light in ("area", "both", "all"):
        y = 1.98
        q = G.quad((-0.25, y, -0.25), (0.25, y, -0.25), (0.25, y, 0.25), (-0.25, y, 0.25), normal=(0, -1, 0))
        s.push(q, MediumInterface(MatteMaterial(Kd=RGBSpectrum(0.0)), emission=Emissive(Le=RGBSpectrum(1.0), scale=1.0, two_sided=False)))
    s.sync()
    film = Film((width, height))
    cam = PerspectiveCamera((0, 1, -3.5), (0, 1, 0), film, fov=40.0)
    return s, film, cam


def integration_test_scene(width=64, height=64, with_fog=True):
    """test/volpath_integration.jl:9-115: matte walls, fog-filled glass sphere, gold conductor sphere, PointLight(15)."""
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
    if with_fog:
        from .media import HomogeneousMedium
        fog = HomogeneousMedium(sigma_a=RGBSpectrum(0.01), sigma_s=RGBSpectrum(0.3), Le=RGBSpectrum(0.0), g=0.3)
        s.push(G.sphere((-0.4, 0.4, 0.0), 0.35, 32), MediumInterface(glass, inside=fog, outside=None))
    else:
        s.push(G.sphere((-0.4, 0.4, 0.0), 0.35, 32), glass)
    s.push(G.sphere((0.4, 0.35, 0.0), 0.3, 32), gold)
    s.push(PointLight((0, 1.8, 0), RGBSpectrum(15.0)))
    s.sync()
    film = Film((width, height))
    cam = PerspectiveCamera((0, 1, -3.5), (0, 1, 0), film, fov=40.0)
    return s, film, cam


def slab_scene(width=32, height=32, medium=None, thickness=1.0, inner_emitter=False):
    """Closed-form check scene: the camera looks through a slab of `medium` (index-matched
    interface: GlassMaterial(Kr=0, Kt=1, index=1)) at a large two-sided emitter; pixel ~ Le * exp(-sigma_t * d).
    inner_emitter: a two-sided emitter INSIDE the medium (the same medium on both of its sides) — the one place where a ray
    scattered by the phase function meets emission without a specular vertex in between, i.e. where r_l = r_u / phase_pdf is used."""
    s = Scene()
    emitter = G.quad((-4, -4, 3), (4, -4, 3), (4, 4, 3), (-4, 4, 3), normal=(0, 0, -1))
    s.push(emitter, MediumInterface(MatteMaterial(Kd=RGBSpectrum(0.0)), emission=Emissive(Le=RGBSpectrum(0.5), scale=1.0, two_sided=True)))
    if medium is not None:
        iface = MediumInterface(GlassMaterial(Kr=RGBSpectrum(0.0), Kt=RGBSpectrum(1.0), index=1.0), inside=medium, outside=None)
        s.push(G.rect3f((-2.5, -2.6, 1.0), (5.0, 5.2, thickness)), iface)
        if inner_emitter:
            z = 1.0 + 0.55 * thickness
            inner = G.quad((-1.3, -1.1, z), (1.1, -1.1, z), (1.1, 1.2, z), (-1.3, 1.2, z), normal=(0, 0, -1))
            s.push(inner, MediumInterface(MatteMaterial(Kd=RGBSpectrum(0.0)), inside=medium, outside=medium,
                                          emission=Emissive(Le=RGBSpectrum(1.0, 0.8, 0.6), scale=1.0, two_sided=True)))
    s.sync()
    film = Film((width, height))
    cam = PerspectiveCamera((0, 0, -2), (0, 0, 1), film, fov=20.0)
    return s, film, cam


def cloud_density(n=(64, 64, 32), seed=7, fill=0.35):
    """Seeded synthetic cloud field (worley-ish blobs, thresholded) standing in for the BOMEX LES data that is
    not in the reference tree."""
    rng = np.random.default_rng(seed)
    nx, ny, nz = n
    x, y, z = np.meshgrid(np.linspace(0, 1, nx), np.linspace(0, 1, ny), np.linspace(0, 1, nz), indexing="ij")
    d = np.zeros(n, dtype=np.float64)
    for _ in range(24):
        c = rng.random(3) * np.array([1, 1, 0.6]) + np.array([0, 0, 0.2])
        r = 0.06 + 0.12 * rng.random()
        d += np.exp(-(((x - c[0]) ** 2 + (y - c[1]) ** 2 + ((z - c[2]) * 1.6) ** 2) / (r * r)))
    d = np.clip(d - np.quantile(d, 1.0 - fill), 0, None)
    d = d / max(d.max(), 1e-9)
    return d.astype(np.float32)


def cloud_scene(width=1024, height=1024, kind="nanovdb", res=(128, 128, 64), sigma_scale=60.0, majorant_res=(32, 32, 32)):
    """Config 4 stand-in: synthetic cloud in a 1.2 cube at (-0.6, 0.3, -0.6), sigma_a = 0, sigma_s = 1,
    g = 0.877, index-matched glass cube as MediumInterface(inside = cloud); Ambient(.03,.07,.23) + Directional(2.6,2.5,2.3)
    (examples/bomex_cloud_example.jl:131-143), matte ground plane; camera looking at the cloud."""
    from .lights import AmbientLight
    from .media import GridMedium, NanoVDBMedium
    dens = cloud_density(res) * np.float32(sigma_scale)
    lo, hi = (-0.6, 0.3, -0.6), (0.6, 1.5, 0.6)
    if kind == "nanovdb":
        med = NanoVDBMedium(dens, bounds=(lo, hi), sigma_a=RGBSpectrum(0.0), sigma_s=RGBSpectrum(1.0), g=0.877, majorant_res=tuple(majorant_res))
    else:
        med = GridMedium(dens, sigma_a=RGBSpectrum(0.0), sigma_s=RGBSpectrum(1.0), g=0.877, bounds=(lo, hi))
    s = Scene()
    s.push(AmbientLight(RGBSpectrum(0.03, 0.07, 0.23)))
    s.push(DirectionalLight(RGBSpectrum(2.6, 2.5, 2.3), (-0.5826, -0.766, -0.2717)))
    s.push(G.rect3f((-4, -0.01, -4), (8, 0.01, 8)), MatteMaterial(Kd=RGBSpectrum(0.35, 0.33, 0.3)))
    eps = 1e-3
    iface = MediumInterface(GlassMaterial(Kr=RGBSpectrum(0.0), Kt=RGBSpectrum(1.0), index=1.0), inside=med, outside=None)
    s.push(G.rect3f((lo[0] - eps, lo[1] - eps, lo[2] - eps), (1.2 + 2 * eps, 1.2 + 2 * eps, 1.2 + 2 * eps)), iface)
    s.sync()
    film = Film((width, height))
    cam = PerspectiveCamera((0.0, 1.0, -3.2), (0.0, 0.85, 0.0), film, fov=35.0)
    return s, film, cam


def bomex_density(res=(256, 256, 128), fill=0.05, max_extinction=620.0, cache=True):
    """Config 4's density field: the worley-fbm recipe of `generate_cloud_density` (src/random.jl:149-206, pure noise:
    `sphere_falloff = false` — a cloud FIELD, not one cloud) on a 256 x 256 x 128 grid, the threshold set to the field's
    (1 - fill) quantile so that `fill` (5 %) of the voxels are cloudy, scaled so that the densest voxel has extinction 620
    (the BOMEX example's `extinction_scale` comment, examples/bomex_cloud_example.jl:50).  The LES data itself is not in the
    reference tree (:22 points outside it).  The noise takes ~10 s of host time at full size, so the array is kept in the temp dir."""
    import hashlib
    import os
    import tempfile
    from . import noise
    res = tuple(int(v) for v in res)
    key = hashlib.sha1(repr(("bomex-v1", res, float(fill), float(max_extinction))).encode()).hexdigest()[:16]
    path = os.path.join(tempfile.gettempdir(), "hikari_mi355x_cache", "bomex_%s.npy" % key)
    if cache and os.path.isfile(path):
        try:
            d = np.load(path)
            if d.shape == res and d.dtype == np.float32:
                return d
        except (OSError, ValueError):
            pass
    base, _ = noise.cloud_noise_base(res, scale=4.0, worley_weight=0.6)
    thr = float(np.quantile(base, 1.0 - fill))
    val = np.clip((base - thr) / (1.0 - thr), 0.0, 1.0)
    d = (val * (max_extinction / max(float(val.max()), 1e-12))).astype(f32)
    if cache:
        try:
            os.makedirs(os.path.dirname(path), exist_ok=True)
            tmp = "%s.%d.tmp.npy" % (path, os.getpid())
            np.save(tmp, d)
            os.replace(tmp, path)
        except OSError:
            pass
    return d


def bomex_scene(width=1024, height=1024, res=(256, 256, 128), fill=0.05, max_extinction=620.0, majorant_res=(64, 64, 64), kind="nanovdb",
                sigma_a=None, sigma_s=None, g=0.877):
    """Config 4 (examples/bomex_cloud_example.jl:53-184 `example_bomex_disney_lighting`): `bomex_density` as a NanoVDB
    medium (`build_nanovdb_from_dense`) in the cube 1.2 at (-0.6, 0.3, -0.6), sigma_a = 0, sigma_s = 1, g = 0.877, 64^3 majorant
    grid; the cube's faces are index-matched glass (Kr = 0, Kt = 1, index = 1) with MediumInterface(inside = cloud); floor, red
    left wall and green right wall, no ceiling / back wall; AmbientLight(.03,.07,.23) + DirectionalLight(2.6,2.5,2.3) along
    (-.5826,-.766,-.2717); camera (0,1,-3.5) -> (0,.9,0), fov 40.  `sigma_a` / `sigma_s` / `g` default to the example's (0, 1, 0.877);
    the tests also render an absorbing variant of the same field."""
    from .lights import AmbientLight
    from .media import GridMedium, NanoVDBMedium
    dens = bomex_density(res, fill, max_extinction)
    lo, hi = (-0.6, 0.3, -0.6), (0.6, 1.5, 0.6)
    sigma_a = RGBSpectrum(0.0) if sigma_a is None else sigma_a
    sigma_s = RGBSpectrum(1.0) if sigma_s is None else sigma_s
    if kind == "nanovdb":
        med = NanoVDBMedium(dens, bounds=(lo, hi), sigma_a=sigma_a, sigma_s=sigma_s, g=g, majorant_res=tuple(majorant_res))
    else:
        med = GridMedium(dens, sigma_a=sigma_a, sigma_s=sigma_s, g=g, bounds=(lo, hi), majorant_res=tuple(majorant_res))
    white = MatteMaterial(Kd=RGBSpectrum(0.73, 0.73, 0.73))
    red = MatteMaterial(Kd=RGBSpectrum(0.65, 0.05, 0.05))
    green = MatteMaterial(Kd=RGBSpectrum(0.12, 0.45, 0.15))
    box, half = 2.0, 1.0
    s = Scene()
    s.push(AmbientLight(RGBSpectrum(0.03, 0.07, 0.23)))
    s.push(DirectionalLight(RGBSpectrum(2.6, 2.5, 2.3), (-0.5826, -0.766, -0.2717)))
    s.push(G.rect3f((-half, 0, -half), (box, 0.01, box)), white)
    s.push(G.rect3f((-half, 0, -half), (0.01, box, box)), red)
    s.push(G.rect3f((half - 0.01, 0, -half), (0.01, box, box)), green)
    iface = MediumInterface(GlassMaterial(Kr=RGBSpectrum(0.0), Kt=RGBSpectrum(1.0), index=1.0), inside=med, outside=None)
    s.push(G.rect3f(lo, (1.2, 1.2, 1.2)), iface)
    s.sync()
    film = Film((width, height))
    cam = PerspectiveCamera((0.0, 1.0, -3.5), (0.0, 0.9, 0.0), film, fov=40.0)
    return s, film, cam


def material_scene(width=64, height=64, material=None, light="both", thin_panel=False):
    """Cornell-like box whose sphere + tilted slab carry `material` (the kinds of Appendix A that the Cornell /
    integration scenes do not exercise: coated / thin / transmissive).  `thin_panel` hangs a single-sided-thin
    quad of the material in front of the back wall so transmission lobes carry light."""
    white = MatteMaterial(Kd=RGBSpectrum(0.73, 0.73, 0.73))
    red = MatteMaterial(Kd=RGBSpectrum(0.65, 0.05, 0.05))
    green = MatteMaterial(Kd=RGBSpectrum(0.12, 0.45, 0.15))
    box, half = 2.0, 1.0
    s = Scene()
    s.push(G.rect3f((-half, 0, -half), (box, 0.01, box)), white)
    s.push(G.rect3f((-half, box - 0.01, -half), (box, 0.01, box)), white)
    s.push(G.rect3f((-half, 0, half - 0.01), (box, box, 0.01)), white)
    s.push(G.rect3f((-half, 0, -half), (0.01, box, box)), red)
    s.push(G.rect3f((half - 0.01, 0, -half), (0.01, box, box)), green)
    s.push(G.sphere((-0.4, 0.4, 0.1), 0.35, 24), material)
    if thin_panel:
        s.push(G.quad((0.05, 0.1, 0.45), (0.85, 0.1, 0.15), (0.85, 1.3, 0.15), (0.05, 1.3, 0.45), normal=(-0.35, 0.0, -0.94)), material)
    else:
        s.push(G.rect3f((0.15, 0.0, -0.1), (0.5, 0.6, 0.5)), material)
    if light in ("point", "both"):
        s.push(PointLight.from_spectrum_first(RGBSpectrum(4.0), (0.3, 1.7, -0.6)))
    if light in ("area", "both"):
        y = 1.98
        q = G.quad((-0.25, y, -0.25), (0.25, y, -0.25), (0.25, y, 0.25), (-0.25, y, 0.25), normal=(0, -1, 0))
        s.push(q, MediumInterface(MatteMaterial(Kd=RGBSpectrum(0.0)), emission=Emissive(Le=RGBSpectrum(6.0), scale=1.0, two_sided=False)))
    s.sync()
    film = Film((width, height))
    cam = PerspectiveCamera((0, 1, -3.5), (0, 1, 0), film, fov=40.0)
    return s, film, cam


def sky_scene(width=800, height=800, env_res=512, gold=None, sun=True, tess=64, analytic=False, intensity=1.0, turbidity=3.0):
    """Config 3 (README.md:62-75): glass sphere r = 1 at the origin, `Gold(roughness=0.01)` slab Rect3f((-2,-2,-1),(4,4,.01)),
    `SunSkyLight(Vec3f(1,2,9); intensity=1, turbidity=3, ground_enabled=false)` = the Hosek-Wilkie bake of lights/sun_sky.jl:
    EnvironmentLight(512^2 equal-area map, scale = intensity/10567) + SunLight(RGB(5,4.75,4.25)*intensity, -dir).
    `analytic=True` swaps the bake for `analytic_sky` (a synthetic map with a hot spot, used by the importance-sampling tests)."""
    from .envmap import EnvironmentLight, EnvironmentMap, analytic_sky
    from .lights import D65_PHOTOMETRIC, SunLight
    from .materials import Gold
    from .sunsky import sunsky_to_envlight
    import numpy as _np
    s = Scene()
    s.push(G.sphere((0, 0, 0), 1.0, tess), GlassMaterial(Kr=RGBSpectrum(1.0), Kt=RGBSpectrum(1.0), index=1.5))
    s.push(G.rect3f((-2, -2, -1), (4, 4, 0.01)), gold if gold is not None else Gold(roughness=0.01))
    sun_dir = (1.0, 2.0, 9.0)
    if analytic:
        inv = float(_np.float32(intensity) / _np.float32(D65_PHOTOMETRIC))
        env_light = EnvironmentLight(EnvironmentMap(analytic_sky(env_res, sun_dir=sun_dir)), RGBSpectrum(inv, inv, inv))
        sun_light = SunLight.from_rgb((5.0 * intensity, 4.75 * intensity, 4.25 * intensity), tuple(-v for v in sun_dir))
    else:
        env_light, sun_light = sunsky_to_envlight(sun_dir, intensity=intensity, turbidity=turbidity, ground_enabled=False, resolution=env_res)
    s.push(env_light)
    if sun:
        s.push(sun_light)
    s.sync()
    film = Film((width, height))
    cam = PerspectiveCamera((4.0, -5.0, 2.5), (0, 0, -0.3), film, up=(0, 0, 1), fov=40.0)
    return s, film, cam


def textured_scene(width=64, height=64):
    """Texture paths of SURVEY row a32 in one box: RGBA Kd texture (bilinear, (1-v, u) flip), alpha cut-outs (stochastic alpha
    test in closest-hit and shadow rays, intersection.jl:142-221, 302-406), Float32 roughness / sigma textures, textured
    CoatedDiffuse reflectance and a textured emitter (diffuse-area.jl:53-82)."""
    from .materials import CoatedDiffuseMaterial, Texture
    rng = np.random.default_rng(3)
    yy, xx = np.mgrid[0:16, 0:16]
    checker = np.where(((xx // 2 + yy // 2) % 2)[..., None] == 0, np.array([0.8, 0.75, 0.2], f32), np.array([0.15, 0.2, 0.7], f32)).astype(f32)
    checker = np.concatenate([checker, np.ones((16, 16, 1), f32)], axis=2)
    leaf = np.zeros((12, 12, 4), f32)
    leaf[..., 0:3] = np.array([0.2, 0.7, 0.25], f32) * (0.6 + 0.4 * rng.random((12, 12, 1)).astype(f32))
    r = np.hypot(xx[:12, :12] - 5.5, yy[:12, :12] - 5.5)
    leaf[..., 3] = np.clip((6.0 - r) / 2.0, 0.0, 1.0)           # opaque disc, soft edge, transparent corners
    rough = (0.02 + 0.5 * rng.random((8, 8))).astype(f32)
    sigma = (40.0 * rng.random((4, 4))).astype(f32)
    refl = rng.random((8, 8, 3)).astype(f32)
    glow = (np.array([1.0, 0.9, 0.7], f32) * (0.3 + 0.7 * rng.random((6, 6, 1)).astype(f32))).astype(f32)
    white = MatteMaterial(Kd=RGBSpectrum(0.73, 0.73, 0.73))
    box, half = 2.0, 1.0
    s = Scene()
    s.push(G.rect3f((-half, 0, -half), (box, 0.01, box)), MatteMaterial(Kd=Texture(checker), sigma=Texture(sigma)))
    s.push(G.rect3f((-half, box - 0.01, -half), (box, 0.01, box)), white)
    from .materials import VertexColorTexture
    back = G.rect3f((-half, 0, half - 0.01), (box, box, 0.01))
    s.push(back, MatteMaterial(Kd=VertexColorTexture(0.2 + 0.7 * rng.random((back.n_faces, 3, 3)).astype(f32))))   # per-face vertex colours
    s.push(G.rect3f((-half, 0, -half), (0.01, box, box)), MatteMaterial(Kd=RGBSpectrum(0.65, 0.05, 0.05)))
    s.push(G.rect3f((half - 0.01, 0, -half), (0.01, box, box)), MatteMaterial(Kd=RGBSpectrum(0.12, 0.45, 0.15)))
    s.push(G.sphere((-0.45, 0.35, 0.1), 0.35, 24), ConductorMaterial(eta=RGBSpectrum(0.2, 0.92, 1.1), k=RGBSpectrum(3.9, 2.45, 2.14), roughness=Texture(rough)))
    s.push(G.rect3f((0.2, 0.0, 0.0), (0.5, 0.6, 0.5)), CoatedDiffuseMaterial(reflectance=Texture(refl), u_roughness=0.1, v_roughness=0.1))
    s.push(G.quad((-0.7, 0.2, -0.5), (0.3, 0.2, -0.6), (0.3, 1.2, -0.6), (-0.7, 1.2, -0.5), normal=(0.1, 0.0, -0.995)), MatteMaterial(Kd=Texture(leaf)))
    y = 1.98
    q = G.quad((-0.3, y, -0.3), (0.3, y, -0.3), (0.3, y, 0.3), (-0.3, y, 0.3), normal=(0, -1, 0))
    s.push(q, MediumInterface(MatteMaterial(Kd=RGBSpectrum(0.0)), emission=Emissive(Le=Texture(glow), scale=6.0, two_sided=False)))
    s.push(PointLight.from_spectrum_first(RGBSpectrum(2.0), (0.5, 1.6, -0.7)))
    s.sync()
    film = Film((width, height))
    cam = PerspectiveCamera((0, 1, -3.5), (0, 1, 0), film, fov=40.0)
    return s, film, cam


def _boxes_mesh(centers, half, angle):
    """n axis-rotated (about y) boxes as one triangle soup, vectorised: 12 triangles per box, flat normals, uv per face."""
    n = centers.shape[0]
    sx = np.array([-1, 1, 1, -1, -1, 1, 1, -1], np.float64)
    sy = np.array([-1, -1, 1, 1, -1, -1, 1, 1], np.float64)
    sz = np.array([-1, -1, -1, -1, 1, 1, 1, 1], np.float64)
    lx, ly, lz = half[:, 0:1] * sx, half[:, 1:2] * sy, half[:, 2:3] * sz          # [n, 8]
    ca, sa = np.cos(angle)[:, None], np.sin(angle)[:, None]
    wx = centers[:, 0:1] + ca * lx + sa * lz
    wy = centers[:, 1:2] + ly
    wz = centers[:, 2:3] - sa * lx + ca * lz
    V = np.stack([wx, wy, wz], axis=-1)                                              # [n, 8, 3]
    quads = np.array([[0, 3, 2, 1], [4, 5, 6, 7], [0, 1, 5, 4], [3, 7, 6, 2], [0, 4, 7, 3], [1, 2, 6, 5]])   # outward CCW
    tris = np.concatenate([quads[:, [0, 1, 2]], quads[:, [0, 2, 3]]], axis=0)       # [12, 3]
    P = V[:, tris, :].reshape(n * 12, 3, 3).astype(f32)
    e1, e2 = P[:, 1] - P[:, 0], P[:, 2] - P[:, 0]
    nrm = np.cross(e1, e2)
    nrm = nrm / np.maximum(np.linalg.norm(nrm, axis=1, keepdims=True), 1e-20)
    N = np.repeat(nrm[:, None, :], 3, axis=1).astype(f32)
    return P, N


def many_light_scene(width=1024, height=1024, n_boxes=83334, emissive_frac=0.05, seed=1, box_scale=1.0):
    """Config 5 stand-in: ~10^6 triangles as boxes in concentric barrel layers around the z axis, 5 % of the boxes
    emissive with a random Le in [0.2, 1] per face (=> ~5*10^4 DiffuseAreaLights in the light BVH), the rest matte / conductor.
    Per-face emission colours come through the reference's own mechanism: a textured Emissive is point-sampled at each face's
    centroid uv (scene-mesh.jl:49), so every face of the emissive mesh carries a uv that selects one texel.  box_scale > 1: the same
    layout with larger boxes — a small case of a few hundred boxes that still fills the frame (the per-pixel pin of the tests)."""
    from .geometry import Mesh
    from .materials import Texture
    rng = np.random.default_rng(seed)
    layers = 12
    layer = rng.integers(0, layers, n_boxes)
    radius = 1.0 + 0.45 * layer + 0.1 * rng.random(n_boxes)
    phi = rng.random(n_boxes) * 2 * np.pi
    zpos = (rng.random(n_boxes) * 2 - 1) * 6.0
    centers = np.stack([radius * np.cos(phi), radius * np.sin(phi), zpos], axis=1)
    half = (0.012 + 0.03 * rng.random((n_boxes, 3))) * box_scale
    P, N = _boxes_mesh(centers, half, rng.random(n_boxes) * np.pi)
    group = rng.random(n_boxes)
    em = group < emissive_frac
    cond = (group >= emissive_frac) & (group < emissive_frac + 0.25)
    matte = ~(em | cond)
    sel = lambda m: np.repeat(m, 12)
    s = Scene()
    s.push(Mesh(P[sel(matte)], N[sel(matte)], None), MatteMaterial(Kd=RGBSpectrum(0.6, 0.6, 0.62)))
    s.push(Mesh(P[sel(cond)], N[sel(cond)], None), ConductorMaterial(eta=RGBSpectrum(0.2, 0.92, 1.1), k=RGBSpectrum(3.9, 2.45, 2.14), roughness=0.2))
    tex_res = 256
    le_tex = (0.2 + 0.8 * rng.random((tex_res, tex_res, 3))).astype(f32)
    n_em = int(em.sum()) * 12
    tu, tv = rng.integers(0, tex_res, n_em), rng.integers(0, tex_res, n_em)
    # nearest lookup idx = trunc(1 + (res - 1) * coord): put the uv at the texel's exact coordinate ((1 - v) selects the row)
    uu = (tu / (tex_res - 1)).astype(f32)
    vv = (1.0 - tv / (tex_res - 1)).astype(f32)
    UV = np.repeat(np.stack([uu, vv], axis=1)[:, None, :], 3, axis=1).astype(f32)
    s.push(Mesh(P[sel(em)], N[sel(em)], UV), MediumInterface(MatteMaterial(Kd=RGBSpectrum(0.0)), emission=Emissive(Le=Texture(le_tex), scale=4.0, two_sided=False)))
    s.sync()
    film = Film((width, height))
    cam = PerspectiveCamera((0.0, -0.2, 9.0), (0.8, 0.3, 0.0), film, up=(0, 1, 0), fov=60.0)
    return s, film, cam

"""Seeded random scenes for the differential parity test (tests/test_fuzz_parity.py): geometry, materials, lights, media, camera,
film size, filter and integrator parameters are all drawn from one numpy Generator, from the vocabulary of SURVEY §8(a).  The
generator only uses the host mirror's public API (the way a user of the reference would write a scene); it never looks at the
oracle or the library."""
import numpy as np

f32 = np.float32


def _rgb(hk, rng, lo=0.05, hi=0.95):
    return hk.RGBSpectrum(*[float(f32(v)) for v in lo + (hi - lo) * rng.random(3)])


def _texture(hk, rng, channels, lo=0.05, hi=0.95):
    h, w = int(rng.integers(2, 9)), int(rng.integers(2, 9))
    shape = (h, w) if channels == 1 else (h, w, channels)
    return hk.Texture((lo + (hi - lo) * rng.random(shape)).astype(f32))


def _maybe_tex(hk, rng, p=0.25):
    return _texture(hk, rng, 3) if rng.random() < p else _rgb(hk, rng)


def closed_form_material(hk, rng, allow_mix=True):
    """A material whose sample / evaluate are closed-form per vertex (strict frame parity at every depth)."""
    k = int(rng.integers(0, 11 if allow_mix else 10))
    if k == 0:
        return hk.MatteMaterial(Kd=_maybe_tex(hk, rng), sigma=float(rng.choice([0.0, 0.0, 20.0, 45.0])))
    if k == 1:
        return hk.MatteMaterial(Kd=_maybe_tex(hk, rng), sigma=_texture(hk, rng, 1, 0.0, 40.0) if rng.random() < 0.3 else 0.0)
    if k == 2:
        return hk.MirrorMaterial(Kr=_rgb(hk, rng, 0.5, 0.95))
    if k == 3:
        rough = float(rng.choice([0.0, 0.0, 0.05, 0.3]))
        return hk.GlassMaterial(Kr=_rgb(hk, rng, 0.6, 1.0), Kt=_rgb(hk, rng, 0.6, 1.0), u_roughness=rough, v_roughness=rough * float(rng.choice([1.0, 0.5])),
                                index=float(rng.choice([1.0, 1.33, 1.5, 2.4])), remap_roughness=bool(rng.integers(0, 2)))
    if k == 4:
        return hk.ConductorMaterial(eta=_rgb(hk, rng, 0.1, 1.5), k=_rgb(hk, rng, 1.5, 4.0), roughness=float(rng.choice([0.0, 0.01, 0.1, 0.4])),
                                    reflectance=_rgb(hk, rng, 0.7, 1.0), remap_roughness=bool(rng.integers(0, 2)))
    if k == 5:
        return [hk.Gold, hk.Silver, hk.Copper, hk.Aluminum, hk.Brass][int(rng.integers(0, 5))](roughness=float(rng.choice([0.0, 0.02, 0.2])))
    if k == 6:
        return hk.ThinDielectricMaterial(eta=float(rng.choice([1.2, 1.5, 1.8])))
    if k == 7:
        return hk.DiffuseTransmissionMaterial(reflectance=_rgb(hk, rng, 0.05, 0.5), transmittance=_rgb(hk, rng, 0.05, 0.5), scale=float(rng.choice([1.0, 0.7])))
    if k == 8:
        r = float(rng.choice([0.0, 0.1, 0.3]))
        kw = dict(interface_u_roughness=float(rng.choice([0.0, 0.05, 0.2])), conductor_u_roughness=r, conductor_v_roughness=r,
                  thickness=float(rng.choice([0.01, 0.05])), albedo=hk.RGBSpectrum(float(rng.choice([0.0, 0.6]))), interface_eta=float(rng.choice([1.3, 1.5])))
        kw["interface_v_roughness"] = kw["interface_u_roughness"]
        if rng.random() < 0.5:
            kw["reflectance"] = _rgb(hk, rng, 0.3, 0.9)
        return hk.CoatedConductorMaterial(**kw)
    if k == 9:
        return hk.MatteMaterial(Kd=_rgb(hk, rng))
    a = closed_form_material(hk, rng, allow_mix=False)
    b = closed_form_material(hk, rng, allow_mix=rng.random() < 0.3)
    amount = _texture(hk, rng, 1, 0.0, 1.0) if rng.random() < 0.3 else float(rng.random())
    return hk.MixMaterial((a, b), amount)


def walk_material(hk, rng):
    """LayeredBxDF kinds (their PCG32 walk is seeded from direction bits: statistical parity beyond the first vertex)."""
    r = float(rng.choice([0.0, 0.1, 0.3]))
    kw = dict(u_roughness=r, v_roughness=r, thickness=float(rng.choice([0.01, 0.1])), eta=float(rng.choice([1.3, 1.5])),
              albedo=hk.RGBSpectrum(float(rng.choice([0.0, 0.5]))), g=float(rng.choice([0.0, 0.4])), max_depth=int(rng.choice([4, 10])),
              n_samples=int(rng.choice([1, 2])))
    if rng.random() < 0.5:
        return hk.CoatedDiffuseMaterial(reflectance=_maybe_tex(hk, rng), **kw)
    return hk.CoatedDiffuseTransmissionMaterial(reflectance=_rgb(hk, rng, 0.05, 0.5), transmittance=_rgb(hk, rng, 0.05, 0.5), **kw)


def _medium(hk, rng, scattering, lo, hi):
    kind = int(rng.integers(0, 4))
    sa = _rgb(hk, rng, 0.05, 1.5)
    ss = _rgb(hk, rng, 0.2, 2.5) if scattering else hk.RGBSpectrum(0.0)
    g = float(rng.choice([0.0, 0.5, -0.3]))
    if kind == 0:
        return hk.HomogeneousMedium(sigma_a=sa, sigma_s=ss, Le=_rgb(hk, rng, 0.0, 0.1) if rng.random() < 0.3 else hk.RGBSpectrum(0.0), g=g)
    res = tuple(int(v) for v in rng.integers(3, 12, 3))
    dens = ((rng.random(res) ** 2) * float(rng.choice([1.0, 4.0]))).astype(f32)
    if rng.random() < 0.4:
        dens[rng.random(res) < 0.5] = 0.0
    mres = tuple(int(v) for v in rng.integers(1, 5, 3))
    if kind == 1:
        return hk.GridMedium(dens, sigma_a=sa, sigma_s=ss, g=g, bounds=(tuple(lo), tuple(hi)), majorant_res=mres)
    if kind == 2:
        # NanoVDB tracks one flat extinction: grey coefficients
        return hk.NanoVDBMedium(dens, (tuple(lo), tuple(hi)), sigma_a=hk.RGBSpectrum(float(rng.choice([0.0, 0.3]))),
                                sigma_s=hk.RGBSpectrum(float(rng.choice([0.5, 2.0])) if scattering else 0.0), g=g, majorant_res=mres)
    sag = (rng.random(res + (3,)) * 1.2).astype(f32)
    ssg = (rng.random(res + (3,)) ** 2 * 2.0).astype(f32) if scattering else np.zeros(res + (3,), f32)
    leg = (rng.random(res + (3,)) * 0.2).astype(f32) if rng.random() < 0.5 else None
    return hk.RGBGridMedium(sigma_a_grid=sag, sigma_s_grid=ssg, Le_grid=leg, sigma_scale=float(rng.choice([1.0, 2.0])), Le_scale=0.5 if leg is not None else 0.0,
                            g=g, bounds=(tuple(lo), tuple(hi)), majorant_res=mres)


def random_scene(hk, seed, klass="closed", size=None):
    """klass: "closed" (closed-form materials, no scattering media: strict parity), "absorbing" (adds absorbing / emitting media:
    strict), "walk" (LayeredBxDF kinds) or "scatter" (scattering media) — the last two compare statistically; "wild" /
    "wild_scatter": the closed / scatter vocabulary plus the awkward cases — a camera INSIDE a medium, a medium nested in a medium,
    coplanar overlapping and zero-area triangles, a swarm of small emitters (a deep light BVH), a rotated environment map, transforms
    on push, deeper paths and larger sample counts.
    Returns (scene, film, camera, integrator keywords, description)."""
    from hikari_jl_amd import geometry as G
    classes = ["closed", "absorbing", "walk", "scatter", "wild", "wild_scatter"]
    rng = np.random.default_rng(1000003 * (classes.index(klass) + 1) + seed)
    wild = klass.startswith("wild")
    if wild:
        klass = "scatter" if klass == "wild_scatter" else ("absorbing" if rng.random() < 0.5 else "closed")
    w, h = int(rng.integers(16, 49)), int(rng.integers(16, 49))
    if size is not None:        # same scene, larger film: enough paths to fill every wave segment of the device several times over
        w, h = size
    s = hk.Scene()
    desc = ["%dx%d" % (w, h)]
    mat = (lambda: walk_material(hk, rng) if rng.random() < 0.6 else closed_form_material(hk, rng)) if klass == "walk" else (lambda: closed_form_material(hk, rng))
    # a room (some walls missing so that rays escape) or an open floor
    half = 1.0
    walls = {"floor": G.rect3f((-half, 0, -half), (2, 0.01, 2)), "ceiling": G.rect3f((-half, 1.99, -half), (2, 0.01, 2)),
             "back": G.rect3f((-half, 0, half - 0.01), (2, 2, 0.01)), "left": G.rect3f((-half, 0, -half), (0.01, 2, 2)),
             "right": G.rect3f((half - 0.01, 0, -half), (0.01, 2, 2))}
    for name, mesh in walls.items():
        if name == "floor" or rng.random() < 0.6:
            m = mat()
            s.push(mesh, m)
            desc.append("%s:%s" % (name, type(m).__name__))
    # objects
    for i in range(int(rng.integers(1, 5))):
        c = np.array([rng.uniform(-0.6, 0.6), rng.uniform(0.2, 1.2), rng.uniform(-0.5, 0.5)])
        shape = int(rng.integers(0, 4))
        m = mat()
        if shape == 0:
            mesh = G.sphere(tuple(c), float(rng.uniform(0.15, 0.4)), int(rng.choice([6, 12, 20])))
        elif shape == 1:
            e = rng.uniform(0.15, 0.5, 3)
            mesh = G.rect3f(tuple(c - e / 2), tuple(e))
        elif shape == 2:
            a, b = rng.normal(size=3), rng.normal(size=3)
            a, b = 0.4 * a / np.linalg.norm(a), 0.4 * b / np.linalg.norm(b)
            mesh = G.quad(tuple(c - a - b), tuple(c + a - b), tuple(c + a + b), tuple(c - a + b))
        else:
            n_t = int(rng.integers(1, 9))
            tris = (c + rng.uniform(-0.35, 0.35, (n_t, 3, 3))).astype(f32)
            mesh = G.Mesh([[tuple(map(float, v)) for v in t] for t in tris])
        s.push(mesh, m)
        desc.append("obj%d:%s" % (shape, type(m).__name__))
    if wild:
        if rng.random() < 0.5:       # coplanar overlapping quads with different materials (equal-t ties), one of them pushed through a transform
            y0 = float(rng.uniform(0.3, 0.9))
            q1 = G.quad((-0.5, y0, -0.4), (0.3, y0, -0.4), (0.3, y0, 0.3), (-0.5, y0, 0.3))
            q2 = G.quad((-0.4, y0, -0.3), (0.6, y0, -0.3), (0.6, y0, 0.4), (-0.4, y0, 0.4))
            s.push(q1, mat())
            s.push(q2.transformed(G.translate((0.0, -0.25, 0.0))), mat(), transform=G.translate((0.0, 0.25, 0.0)))
            desc.append("coplanar")
        if rng.random() < 0.5:       # zero-area and needle triangles
            a = rng.uniform(-0.5, 0.5, 3) + np.array([0, 0.8, 0])
            b = a + rng.uniform(-0.3, 0.3, 3)
            tris = [[a, a, b], [a, b, (a + b) / 2], [a, b, b + np.array([1e-6, 0, 0])]]
            s.push(G.Mesh([[tuple(map(float, v)) for v in t] for t in tris]), mat())
            desc.append("degenerate")
        if rng.random() < 0.5:       # many small emitters
            n_e = int(rng.choice([40, 150, 400]))
            c = np.stack([rng.uniform(-0.9, 0.9, n_e), rng.uniform(0.05, 1.9, n_e), rng.uniform(-0.9, 0.9, n_e)], 1)
            d1, d2 = rng.normal(size=(n_e, 3)) * 0.02, rng.normal(size=(n_e, 3)) * 0.02
            tris = np.stack([c, c + d1, c + d2], 1).astype(f32)
            s.push(G.Mesh(tris), hk.MediumInterface(hk.MatteMaterial(Kd=_rgb(hk, rng)), emission=hk.Emissive(Le=_rgb(hk, rng, 0.2, 1.0), scale=float(rng.uniform(20, 200)),
                                                                                                   two_sided=bool(rng.integers(0, 2)))))
            desc.append("emitters%d" % n_e)
    # an alpha cut-out panel now and then
    if rng.random() < 0.25:
        rgba = np.concatenate([rng.random((6, 6, 3)), (rng.random((6, 6, 1)) > 0.4).astype(float) * rng.choice([1.0, 0.6])], axis=2).astype(f32)
        s.push(G.quad((-0.6, 0.3, -0.7), (0.2, 0.3, -0.75), (0.2, 1.1, -0.75), (-0.6, 1.1, -0.7)), hk.MatteMaterial(Kd=hk.Texture(rgba)))
        desc.append("alpha")
    # media
    if klass in ("absorbing", "scatter"):
        lo = np.array([rng.uniform(-0.8, -0.2), rng.uniform(0.1, 0.5), rng.uniform(-0.6, -0.1)])
        hi = lo + rng.uniform(0.5, 1.0, 3)
        med = _medium(hk, rng, klass == "scatter", lo, hi)
        boundary = hk.GlassMaterial(Kr=hk.RGBSpectrum(0.0), Kt=hk.RGBSpectrum(1.0), index=1.0) if rng.random() < 0.7 else \
            hk.GlassMaterial(Kr=hk.RGBSpectrum(1.0), Kt=hk.RGBSpectrum(1.0), index=1.33)
        outer = None
        if wild and rng.random() < 0.5:      # the whole room, camera included, sits in a thin homogeneous medium; the box medium is nested in it
            outer = hk.HomogeneousMedium(sigma_a=_rgb(hk, rng, 0.01, 0.15), sigma_s=_rgb(hk, rng, 0.02, 0.2) if klass == "scatter" else hk.RGBSpectrum(0.0),
                                         g=float(rng.choice([0.0, 0.6])))
            s.push(G.rect3f((-3.0, -1.0, -5.0), (6.0, 5.0, 8.0)), hk.MediumInterface(hk.GlassMaterial(Kr=hk.RGBSpectrum(0.0), Kt=hk.RGBSpectrum(1.0), index=1.0),
                                                                                  inside=outer, outside=None))
            desc.append("camera-in-medium")
        s.push(G.rect3f(tuple(lo), tuple(hi - lo)), hk.MediumInterface(boundary, inside=med, outside=outer))
        desc.append("medium:%s" % type(med).__name__)
    # lights: at least one
    n_l = 0
    while n_l == 0:
        if rng.random() < 0.5:
            s.push(hk.PointLight((rng.uniform(-0.7, 0.7), rng.uniform(1.2, 1.9), rng.uniform(-0.8, 0.3)), _rgb(hk, rng, 2.0, 12.0)))
            n_l += 1
            desc.append("point")
        if rng.random() < 0.3:
            s.push(hk.SpotLight((rng.uniform(-0.7, 0.7), 1.8, rng.uniform(-0.8, 0.0)), (rng.uniform(-0.3, 0.3), 0.0, rng.uniform(-0.3, 0.3)), _rgb(hk, rng, 10.0, 30.0),
                                float(rng.uniform(25, 60)), float(rng.uniform(5, 24))))
            n_l += 1
            desc.append("spot")
        if rng.random() < 0.3:
            s.push(hk.DirectionalLight(_rgb(hk, rng, 0.5, 3.0), (rng.uniform(-0.5, 0.5), -1.0, rng.uniform(-0.2, 0.8))))
            n_l += 1
            desc.append("directional")
        if rng.random() < 0.25:
            s.push(hk.SunLight.from_rgb(tuple(rng.uniform(1.0, 5.0, 3)), (rng.uniform(-0.5, 0.5), -1.0, rng.uniform(0.0, 0.8))))
            n_l += 1
            desc.append("sun")
        if rng.random() < 0.25:
            s.push(hk.AmbientLight(_rgb(hk, rng, 0.1, 0.6)))
            n_l += 1
            desc.append("ambient")
        if rng.random() < 0.25:
            from hikari_jl_amd.envmap import EnvironmentLight, EnvironmentMap
            res = int(rng.choice([4, 8, 16]))
            data = (rng.random((res, res, 3)) ** 3 * 2.0).astype(f32)
            rot = None
            if wild and rng.random() < 0.6:
                from hikari_jl_amd.envmap import rotation_matrix
                rot = rotation_matrix(float(rng.uniform(0, 360)), tuple(rng.normal(size=3)))
            s.push(EnvironmentLight(EnvironmentMap(data, rot), hk.RGBSpectrum(float(rng.choice([0.5, 1.0])))))
            n_l += 1
            desc.append("env%d" % res)
        if rng.random() < 0.5:
            y = float(rng.uniform(1.5, 1.95))
            e = float(rng.uniform(0.1, 0.4))
            cx, cz = rng.uniform(-0.4, 0.4, 2)
            q = G.quad((cx - e, y, cz - e), (cx + e, y, cz - e), (cx + e, y, cz + e), (cx - e, y, cz + e), normal=(0, -1, 0))
            Le = _texture(hk, rng, 3, 0.2, 1.0) if rng.random() < 0.2 else _rgb(hk, rng, 0.3, 1.0)
            s.push(q, hk.MediumInterface(hk.MatteMaterial(Kd=hk.RGBSpectrum(0.0)), emission=hk.Emissive(Le=Le, scale=float(rng.uniform(2, 10)), two_sided=bool(rng.integers(0, 2)))))
            n_l += 1
            desc.append("area")
    s.sync()
    film = hk.Film((w, h))
    eye = (rng.uniform(-0.5, 0.5), rng.uniform(0.6, 1.4), rng.uniform(-3.6, -2.2))
    lens = float(rng.choice([0.0, 0.0, 0.0, 0.05]))
    aspect = w / h
    cam = hk.PerspectiveCamera(eye, (rng.uniform(-0.2, 0.2), rng.uniform(0.7, 1.1), 0.0), film, fov=float(rng.uniform(25, 60)), lens_radius=lens,
                               focal_distance=float(rng.uniform(2.5, 4.0)) if lens > 0 else 1e6,
                               screen_window=((-aspect, -1), (aspect, 1)) if rng.random() < 0.5 else ((-1, -1), (1, 1)))
    filt = [None, hk.BoxFilter(), hk.TriangleFilter(), hk.MitchellFilter(), hk.LanczosSincFilter(), hk.GaussianFilter()][int(rng.integers(0, 6))]
    kw = dict(max_depth=int(rng.integers(1, 8)), samples=int(rng.choice([1, 2, 3, 4, 5, 8, 16])), regularize=bool(rng.integers(0, 2)),
              material_coherence=str(rng.choice(["none", "sorted", "per_type"])), max_component_value=float(rng.choice([10.0, 10.0, 2.0])))
    if wild:
        kw["max_depth"] = int(rng.integers(1, 14))
        kw["samples"] = int(rng.choice([1, 6, 7, 12, 24, 32]))
        kw["samples_per_pass"] = int(rng.choice([0, 0, 1, 3, 5]))
        kw["russian_roulette_depth"] = int(rng.choice([3, 1, 8]))
    if filt is not None:
        kw["filter"] = filt
    if rng.random() < 0.2:
        kw["accumulation_eltype"] = "Float64"
    desc.append("depth%d spp%d %s" % (kw["max_depth"], kw["samples"], type(filt).__name__))
    return s, film, cam, kw, " ".join(desc)

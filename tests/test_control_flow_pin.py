"""The wavefront control flow, per pixel, from a second source: tests/ref_volpath_np.py restates K1 - K13 for opaque
matte / mirror / glass surfaces under diffuse area, point, spot, directional and ambient lights in float32 NumPy from the Julia text (own ZSobol, own light BVH, brute-force float64
intersection, no queues) and is compared PIXEL BY PIXEL with the oracle's frame of the Cornell box of BASELINE.json configs[1]: the same
sample indices, the same path per (pixel, sample) — a wrong MIS weight, a wrong dimension of a Sobol draw, a roulette applied one bounce
early, a light pmf taken at the wrong point would move every pixel.  What may differ: roundings (the restatement intersects in float64,
sums in another order) and the few paths a knife-edge decision sends elsewhere (a hit on a shared triangle edge, u within an ulp of a
child probability of the light BVH)."""
import numpy as np
import pytest

import ref_volpath_np as R


def _both(hk, oracle, s, cam, w, h, spp, depth, hits32=False):
    p = hk.integrator_params(max_depth=depth, samples=spp, filter=hk.BoxFilter())
    osc = oracle.OracleScene(s)
    acc, _ = osc.render(p, cam, w, h, spp)
    osc.close()
    ref = oracle.finalize(acc, w, h)
    img, _, _ = R.render(s.desc, cam.record(), hk.tables.load(), w, h, spp, depth, max_component_value=float(p.max_component_value), sobol_spp=spp, hits32=hits32)
    return ref, img


@pytest.mark.parametrize("objects,depth,spp", [("sphere_box", 5, 4), ("two_spheres", 3, 2), (None, 8, 4)])
def test_cornell_frame_per_pixel_against_the_numpy_restatement(hk, oracle, objects, depth, spp):
    from hikari_jl_amd import scenes
    w = h = 32
    if objects is None:
        s, film, cam = scenes.cornell_box(w, h, light="area", spheres=False)
    else:
        s, film, cam = scenes.cornell_box(w, h, light="area", objects=objects)
    ref, img = _both(hk, oracle, s, cam, w, h, spp, depth)
    assert ref.shape == img.shape and np.isfinite(img).all() and ref.max() > 0
    num = np.sqrt(((img - ref) ** 2).sum(axis=2))
    den = np.sqrt((ref ** 2).sum(axis=2)) + 1e-6
    rel = num / den
    print("pixels within 1e-4: %.4f, within 1e-2: %.4f, worst %.3g, mean ratio %.6f" % ((rel <= 1e-4).mean(), (rel <= 1e-2).mean(), rel.max(), img.mean() / ref.mean()))
    # measured: every one of the 1 024 pixels within 7e-5 (1.5e-5 at depth 8 with roulette) in all three scenes
    assert (rel <= 2e-4).mean() >= 0.99            # the same path, to rounding, in (nearly) every pixel
    assert (rel <= 1e-2).mean() >= 0.995           # the rest: a sample of the pixel took another way at a knife edge
    assert abs(img.mean() / ref.mean() - 1.0) < 1e-3


@pytest.mark.parametrize("light,depth,spp", [("point", 5, 4), ("both", 6, 4), ("dir", 4, 4), ("spot", 5, 4), ("ambient", 5, 4), ("all", 6, 4)])
def test_delta_lights_per_pixel_against_the_numpy_restatement(hk, oracle, light, depth, spp):
    """A POINT light in the box, alone and beside the area light: a delta light in the light BVH (a point as bounds, a cone of the whole
    sphere: light-bounds.jl:234-246), its intensity uplifted as an illuminant (polynomial of rgb / 2 max times D65, uplift.jl:515-540),
    Li = scale I / r^2 with pdf 1, and next-event estimation WITHOUT the BSDF's pdf in the MIS weight (lights.jl:583-589) — while the area
    light beside it keeps its MIS and its pmf in the same tree.  And a DIRECTIONAL light shining into the open front ("dir"; "all": with the
    other two): a light without bounds, i.e. an INFINITE light of the sampler — chosen with p_inf = n_inf / (n_inf + 1), the rest of the
    random number remapped for the tree, every tree pmf (next-event estimation and the MIS weight of emission that is hit) scaled by
    1 - p_inf (bvh-light-sampler.jl:105-232); wi against its direction, the shadow ray 10^6 long (lights.jl:108-125).  And a SPOT light
    ("spot"; also part of "all"): -wi taken to the light's frame, nothing outside the cone, the fourth-power edge between the two cosines
    (lights.jl:66-100), its bounds a point with the cone of the spot (light-bounds.jl:248-272).  And an AMBIENT light ("ambient"; part of
    "all"): an infinite light that is NOT a delta light — a uniform direction of the sphere with pdf 1 / 4 pi and the BSDF's pdf in the MIS
    weight (lights.jl:199-221) — and the one kind an ESCAPED ray meets (K7, intersection.jl:622-668: its radiance over average(r_u), or
    over average(r_u + r_l / n_lights * 0) after a non-specular bounce: only an environment map has a pdf there)."""
    from hikari_jl_amd import scenes
    w = h = 32
    s, film, cam = scenes.cornell_box(w, h, light=light)
    ref, img = _both(hk, oracle, s, cam, w, h, spp, depth)
    assert ref.shape == img.shape and np.isfinite(img).all() and ref.max() > 0
    rel = np.sqrt(((img - ref) ** 2).sum(axis=2)) / (np.sqrt((ref ** 2).sum(axis=2)) + 1e-6)
    print("%s: pixels within 2e-4: %.4f, within 1e-2: %.4f, worst %.3g, mean ratio %.6f" % (light, (rel <= 2e-4).mean(), (rel <= 1e-2).mean(), rel.max(), img.mean() / ref.mean()))
    assert (rel <= 2e-4).mean() >= 0.99 and (rel <= 1e-2).mean() >= 0.995
    assert abs(img.mean() / ref.mean() - 1.0) < 1e-3


@pytest.mark.parametrize("which", ["mirror", "glass"])
def test_specular_objects_per_pixel_against_the_numpy_restatement(hk, oracle, which):
    """The box with a sphere and a slab of MirrorMaterial / GlassMaterial(index 1.5) under the area light, depth 7: the delta lobes
    (throughput without cosine / pdf, r_l = r_u), emission found after a specular bounce taken WITHOUT MIS, no next-event estimation at a
    specular vertex, Fresnel-weighted choice between reflection and refraction, total internal reflection, roulette from depth 4."""
    from hikari_jl_amd import scenes
    w = h = 32
    m = hk.MirrorMaterial(Kr=hk.RGBSpectrum(0.9, 0.8, 0.7)) if which == "mirror" else hk.GlassMaterial(Kr=hk.RGBSpectrum(1.0), Kt=hk.RGBSpectrum(0.95, 1.0, 0.9), index=1.5)
    s, film, cam = scenes.material_scene(w, h, m, light="area")
    ref, img = _both(hk, oracle, s, cam, w, h, 4, 7)
    rel = np.sqrt(((img - ref) ** 2).sum(axis=2)) / (np.sqrt((ref ** 2).sum(axis=2)) + 1e-6)
    print("%s: pixels within 2e-4: %.4f, within 1e-2: %.4f, worst %.3g, mean ratio %.6f" % (which, (rel <= 2e-4).mean(), (rel <= 1e-2).mean(), rel.max(), img.mean() / ref.mean()))
    assert (rel <= 2e-4).mean() >= 0.98 and (rel <= 1e-2).mean() >= 0.99
    assert abs(img.mean() / ref.mean() - 1.0) < 5e-3


@pytest.mark.parametrize("metal,roughness,depth,spp", [("Copper", 0.3, 5, 4), ("Gold", 0.04, 6, 4), ("Silver", 0.0, 5, 2)])
def test_conductor_per_pixel_against_the_numpy_restatement(hk, oracle, metal, roughness, depth, spp):
    """r_l AFTER A ROUGH, NON-MATTE BOUNCE has a second per-pixel source: the Cornell box with a metal sphere — the
    Trowbridge-Reitz lobe of the Conductor (spectral-eval.jl:223-318 sampled through visible normals, :415-486 evaluated for next-event
    estimation, :3667-3861 Fresnel / D / G / pdf), measured eta / k as PiecewiseLinearSpectrum.  Copper at roughness 0.3 (alpha 0.55: the
    plain rough lobe, beta f cos / pdf and r_l = r_u / pdf into the next vertex's emission MIS), Gold at roughness 0.04 (alpha 0.2: a
    path that has had a non-specular bounce samples the REGULARISED lobe, alpha 0.3, while next-event estimation still evaluates the
    sharp one — reflection/microfacet.jl:97-99, surface-eval.jl:425) and Silver at roughness 0 (the effectively smooth branch: a specular
    sample with f = F / cos, Q21)."""
    from hikari_jl_amd import scenes
    w = h = 32
    s, film, cam = scenes.cornell_box(w, h, light="area", object_material=getattr(hk, metal)(roughness=roughness))
    ref, img = _both(hk, oracle, s, cam, w, h, spp, depth)
    assert np.isfinite(img).all() and ref.max() > 0
    rel = np.sqrt(((img - ref) ** 2).sum(axis=2)) / (np.sqrt((ref ** 2).sum(axis=2)) + 1e-6)
    print("%s %.2f: pixels within 2e-4: %.4f, within 1e-2: %.4f, worst %.3g, mean ratio %.6f" % (metal, roughness, (rel <= 2e-4).mean(), (rel <= 1e-2).mean(), rel.max(), img.mean() / ref.mean()))
    assert (rel <= 2e-4).mean() >= 0.98 and (rel <= 1e-2).mean() >= 0.99
    assert abs(img.mean() / ref.mean() - 1.0) < 5e-3


@pytest.mark.parametrize("which", ["scattering", "absorbing", "grid"])
def test_homogeneous_medium_against_the_numpy_restatement(hk, oracle, which):
    """K4 - K6, K10 and K14 have a second source: a HomogeneousMedium behind an index-matched boundary, restated in
    tests/ref_volpath_np.py from delta-tracking.jl:28-58, 154-453 (the LCG seeded by ray bits, absorption / real / null collisions,
    beta and r_u rescaled by T_maj sigma / pdf), medium-scatter.jl:15-198 (next-event estimation with the phase function, the continuation
    with r_l = r_u / phase_pdf, prev_n = wo), intersection.jl:303-542 (shadow rays through medium-transition surfaces, ratio tracking
    with PCG32) and :690-735 (the camera's medium).  The trackers' random streams are seeded by HASHES OF FLOAT BIT PATTERNS, so two
    implementations that differ by one rounding anywhere upstream of a medium boundary draw different (equally valid) streams: a
    per-pixel comparison of single samples is impossible by construction.  What is compared is the CONVERGED estimate: 128 - 192 spp on an
    8 x 8 film in 8 batches each side, channel means within 1 % + 4 standard errors, per-pixel z-scores from the batch variances.
    This is the END-TO-END check of the restated control flow (which medium a ray is in, who traces which shadow ray, what a survivor
    carries to its surface); it is blind to some weights — dropping the division in r_l = r_u / phase_pdf moves these means by 0.1 % —
    which is what the two per-ray tests below are for."""
    from hikari_jl_amd import scenes
    w = h = 8
    if which == "scattering":
        med = hk.HomogeneousMedium(sigma_a=hk.RGBSpectrum(0.2, 0.3, 0.1), sigma_s=hk.RGBSpectrum(0.8, 0.6, 0.9), Le=hk.RGBSpectrum(0.05, 0.0, 0.0), g=0.4)
        depth, batches, per = 6, 8, 24
    elif which == "grid":      # a HETEROGENEOUS medium end to end: the DDA over the majorant cells and the trilinear density inside the restated loop
        gr = np.random.default_rng(8)
        dens = (gr.random((10, 8, 6)) ** 2 * 2.5).astype(np.float32)
        med = hk.GridMedium(dens, sigma_a=hk.RGBSpectrum(0.15, 0.2, 0.1), sigma_s=hk.RGBSpectrum(1.2, 1.0, 1.4), g=0.35, bounds=((-2.5, -2.6, 1.0), (2.5, 2.6, 2.0)), majorant_res=(4, 3, 5))
        depth, batches, per = 6, 8, 16
    else:
        med = hk.HomogeneousMedium(sigma_a=hk.RGBSpectrum(0.7, 0.9, 1.2), sigma_s=hk.RGBSpectrum(0.0), Le=hk.RGBSpectrum(0.0))
        depth, batches, per = 4, 8, 16
    s, film, cam = scenes.slab_scene(w, h, med)
    n = batches * per
    p = hk.integrator_params(max_depth=depth, samples=n, filter=hk.BoxFilter())
    osc = oracle.OracleScene(s)
    A, B = [], []
    for b in range(batches):
        acc, _ = osc.render(p, cam, w, h, per, first=1 + b * per)
        A.append(oracle.finalize(acc, w, h))
        img, _, _ = R.render(s.desc, cam.record(), hk.tables.load(), w, h, per, depth, max_component_value=float(p.max_component_value), first=1 + b * per, sobol_spp=n)
        B.append(img)
    osc.close()
    A, B = np.stack(A), np.stack(B)
    mA, mB = A.mean(0), B.mean(0)
    seA, seB = A.std(0, ddof=1) / np.sqrt(batches), B.std(0, ddof=1) / np.sqrt(batches)
    assert np.isfinite(mB).all() and mA.mean() > 0.1
    for c in range(3):
        a, b_ = mA[..., c].mean(), mB[..., c].mean()
        se = np.sqrt((seA[..., c] ** 2).sum() + (seB[..., c] ** 2).sum()) / (w * h)
        print("%s channel %d: oracle %.4f restatement %.4f (se %.4f)" % (which, c, a, b_, se))
        assert abs(a - b_) <= 0.01 * a + 4.0 * se, (which, c, a, b_, se)
    z = (mA - mB) / np.sqrt(seA ** 2 + seB ** 2 + 1e-12)
    lit = mA > 1e-3
    print("%s: |z| > 4 on %.4f of the lit pixel channels, worst %.2f" % (which, (np.abs(z[lit]) > 4).mean(), np.abs(z[lit]).max()))
    assert (np.abs(z[lit]) > 4).mean() <= 0.03 and np.abs(z[lit]).max() < 8.0


@pytest.mark.parametrize("which", ["homogeneous", "grid", "nanovdb", "rgbgrid"])
def test_shadow_walk_per_ray_against_the_numpy_restatement(hk, oracle, which):
    """K10 ray by ray, bit for bit: the oracle's trace_shadow_transmittance (its test entry hko_medium mode 2) against
    ref_volpath_np.trace_shadow on the SAME rays — the <= 10-segment walk through medium-transition surfaces, the medium on either side
    by the geometric normal, ratio tracking in a homogeneous medium with PCG32 seeded by pbrt_hash(origin), pbrt_hash(direction)
    (MurmurHash64A of the float bits), Russian roulette of the transmittance estimate, the three running products T_ray / r_u / r_l.
    Same inputs, so the hashed seeds agree and the comparison is exact: every visibility decision equal, every zero equal, the products
    within 4 ulp (float32 exp / log of glibc against correctly rounded ones) on all but a few rays per thousand whose tracker crosses
    a segment end on a rounding.  "grid": the same walk through a HETEROGENEOUS GridMedium (10 x 8 x 6 voxels, 4 x 3 x 5 majorant cells):
    ratio tracking cell by cell along the DDA (_ratio_tracking_dda, intersection.jl:446-542), one PCG32 stream across the cells, the
    trilinear density at origin + dir t of every step."""
    from hikari_jl_amd import scenes
    if which == "homogeneous":
        med = hk.HomogeneousMedium(sigma_a=hk.RGBSpectrum(0.2, 0.3, 0.1), sigma_s=hk.RGBSpectrum(0.8, 0.6, 0.9), Le=hk.RGBSpectrum(0.0), g=0.4)
    else:
        gr = np.random.default_rng(8)
        dens = (gr.random((10, 8, 6)) ** 2 * 2.5).astype(np.float32)
        dens[:3, :, :2] = 0.0
        if which == "rgbgrid":
            ga = (gr.random((7, 6, 5, 3)) * 0.5).astype(np.float32)
            gs = (gr.random((7, 6, 5, 3)) * 1.5).astype(np.float32)
            gs[:2, :, :2] = 0.0
            ga[:2, :, :2] = 0.0
            med = hk.RGBGridMedium(sigma_a_grid=ga, sigma_s_grid=gs, sigma_scale=1.7, g=-0.2, bounds=((-2.5, -2.6, 1.0), (2.5, 2.6, 2.0)), majorant_res=(3, 4, 2))
        elif which == "nanovdb":     # (24 x 17 x 11 voxels: leaves of 8^3 that are partly empty, several leaves along every axis)
            big = (gr.random((24, 17, 11)) ** 2 * 2.5).astype(np.float32)
            big[:7, :, :4] = 0.0
            med = hk.NanoVDBMedium(big, bounds=((-2.5, -2.6, 1.0), (2.5, 2.6, 2.0)), sigma_a=hk.RGBSpectrum(0.3, 0.4, 0.2),
                                   sigma_s=hk.RGBSpectrum(1.2, 1.0, 1.4), g=0.35, majorant_res=(4, 3, 5))
        else:
            med = hk.GridMedium(dens, sigma_a=hk.RGBSpectrum(0.3, 0.4, 0.2), sigma_s=hk.RGBSpectrum(1.2, 1.0, 1.4), g=0.35, bounds=((-2.5, -2.6, 1.0), (2.5, 2.6, 2.0)), majorant_res=(4, 3, 5))
    s, _, _ = scenes.slab_scene(16, 16, med, inner_emitter=True)
    tb = R.Tables(hk.tables.load())
    sc = R.SceneNP(s.desc, tb)
    rng = np.random.default_rng(77)
    n = 600
    o = (rng.random((n, 3)) * np.array([6.0, 6.0, 5.0]) + np.array([-3.0, -3.0, -1.5])).astype(np.float32)
    d = rng.normal(size=(n, 3))
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    d[:100] = np.array([0, 0, 1], np.float32)                        # straight through the slab towards the emitter
    tmax = np.where(rng.random(n) < 0.3, 1.0e6, rng.random(n) * 6.0).astype(np.float32)
    lam = (400.0 + 300.0 * rng.random((n, 4))).astype(np.float32)
    inside = (np.abs(o[:, 0]) < 2.5) & (o[:, 1] > -2.6) & (o[:, 1] < 2.6) & (o[:, 2] > 1.0) & (o[:, 2] < 2.0)
    osc = oracle.OracleScene(s)
    ref = np.zeros((n, 13), np.float32)
    for m in (-1, 0):
        sel = np.nonzero(inside == (m == 0))[0]
        ref[sel] = osc.medium(2, m, o[sel], lam[sel], b=d[sel], tmax=tmax[sel])
    osc.close()
    got = np.zeros((n, 13), np.float32)
    for i in range(n):
        T, ru, rl, vis = R.trace_shadow(sc, o[i], d[i], tmax[i], lam[i], 0 if inside[i] else -1)
        got[i] = np.concatenate([T, ru, rl, [1.0 if vis else 0.0]])
    # (a homogeneous medium's majorant is its extinction: the first component of T_ray is 0 or 1, the others carry the spectral ratios)
    assert 0.2 < ref[:, 12].mean() < 0.95 and (ref[:, 0] == 0).mean() > (0.1 if which == "homogeneous" else 0.03) and ((ref[:, 1] > 0) & (ref[:, 1] != 1)).mean() > 0.05
    assert np.array_equal(got[:, 12], ref[:, 12])                                        # visible / blocked: every ray
    ulp = np.abs(got[:, :12].view(np.int32).astype(np.int64) - ref[:, :12].view(np.int32).astype(np.int64))
    same = (ulp <= 4).all(axis=1)
    print("shadow walk: %d rays, %.4f within 4 ulp in all twelve values, zero pattern equal on %.4f" % (n, same.mean(), ((got[:, :4] == 0) == (ref[:, :4] == 0)).all(axis=1).mean()))
    assert same.mean() >= 0.99


@pytest.mark.parametrize("which", ["homogeneous", "grid", "grid_rotated", "nanovdb", "rgbgrid"])
def test_media_stage_per_ray_against_the_numpy_restatement(hk, oracle, which):
    """K4 + K5 + K6 ray by ray: the oracle's own stage code (process_media_stage, reached through its test entry hko_media_stage — nothing
    restated on that side) against ref_volpath_np.media_vertex on the SAME rays, throughputs, weights and Sobol draws.  Same ray bits, so
    both sides seed the LCG alike (delta-tracking.jl:28-58) and the comparison is exact: the FATE of every ray (absorbed / dropped at the
    depth limit / scattered / passed on to its surface / escaped), and for the survivors beta, r_u, r_l after the T_maj rescaling; for a
    scattering vertex the continuation (origin = the collision point, direction from the phase function, beta and r_u unchanged,
    r_l = r_u / phase_pdf — medium-scatter.jl:172-198) and the shadow ray of its next-event estimation (Ld = beta * phase * Li, r_u * phase
    pdf, r_u * light pdf * pmf of the light tree walked WITHOUT a normal, t_max 10^6 for an area light — :15-118); and the emission a
    tentative collision adds to the pixel (delta-tracking.jl:371-381).  A wrong r_l after a phase sample, which the converged test above
    cannot see (measured: dropping the division changes its means by 0.1 %), fails here on every scattered ray.
    "grid": a HETEROGENEOUS medium — BASELINE configs[3]'s kind — through the same comparison: a GridMedium of 10 x 8 x 6 voxels under a
    4 x 3 x 5 majorant grid (cells that do not divide the voxels), i.e. the DDA over the majorant cells (create_dda_iterator / dda_next,
    media.jl:268-500), the tracking state carried from cell to cell with ONE LCG stream, the trilinear density at every tentative collision
    (sample_density, :1544-1595) and the majorant grid itself (build_majorant_grid, :1459-1496: rebuilt from the text and compared with the
    scene description's); "grid_rotated": the same with a medium-to-render transform (rays and points taken to medium space)."""
    from hikari_jl_amd import scenes
    if which == "homogeneous":
        med = hk.HomogeneousMedium(sigma_a=hk.RGBSpectrum(0.2, 0.3, 0.1), sigma_s=hk.RGBSpectrum(0.8, 0.6, 0.9), Le=hk.RGBSpectrum(0.05, 0.02, 0.0), g=0.4)
    else:
        gr = np.random.default_rng(8)
        dens = (gr.random((10, 8, 6)) ** 2 * 2.5).astype(np.float32)
        dens[:3, :, :2] = 0.0                                         # empty majorant cells on the way
        xf = None
        if which == "grid_rotated":
            c, sn = np.cos(0.3), np.sin(0.3)
            xf = np.array([[c, -sn, 0, 0.2], [sn, c, 0, -0.1], [0, 0, 1, 0.05], [0, 0, 0, 1]], np.float32)
        if which == "rgbgrid":     # RGB voxels for sigma_a, sigma_s and Le (emission!), uplifted at every tentative collision (media.jl:1002-1435)
            ga = (gr.random((7, 6, 5, 3)) * 0.5).astype(np.float32)
            gs = (gr.random((7, 6, 5, 3)) * 1.5).astype(np.float32)
            gl = (gr.random((7, 6, 5, 3)) * 0.2).astype(np.float32)
            gs[:2, :, :2] = 0.0
            ga[:2, :, :2] = 0.0
            med = hk.RGBGridMedium(sigma_a_grid=ga, sigma_s_grid=gs, Le_grid=gl, sigma_scale=1.7, Le_scale=0.8, g=-0.2, bounds=((-2.5, -2.6, 1.0), (2.5, 2.6, 2.0)), majorant_res=(3, 4, 2))
        elif which == "nanovdb":     # BASELINE configs[3]'s medium: the NanoVDB tree decoded from its bytes (nanovdb.jl:296-386), the index-space
            # trilinear sampler (:424-470), the world-space majorant grid rebuilt from the tree (:1174-1233) — all restated in MediumNP
            big = (gr.random((24, 17, 11)) ** 2 * 2.5).astype(np.float32)      # (leaves of 8^3 that are partly empty, a second and third leaf along the axes)
            big[:7, :, :4] = 0.0
            med = hk.NanoVDBMedium(big, bounds=((-2.5, -2.6, 1.0), (2.5, 2.6, 2.0)), sigma_a=hk.RGBSpectrum(0.3, 0.4, 0.2),
                                   sigma_s=hk.RGBSpectrum(1.2, 1.0, 1.4), g=0.35, majorant_res=(4, 3, 5))
        else:
            med = hk.GridMedium(dens, sigma_a=hk.RGBSpectrum(0.3, 0.4, 0.2), sigma_s=hk.RGBSpectrum(1.2, 1.0, 1.4), g=0.35, bounds=((-2.5, -2.6, 1.0), (2.5, 2.6, 2.0)),
                                transform=xf, majorant_res=(4, 3, 5))
    s, _, _ = scenes.slab_scene(16, 16, med, inner_emitter=True)
    sc = R.SceneNP(s.desc, R.Tables(hk.tables.load()))
    rng = np.random.default_rng(5)
    n, depth, max_depth = 500, 2, 6
    o = (rng.random((n, 3)) * np.array([4, 4, 0.9]) + np.array([-2, -2, 1.05])).astype(np.float32)
    d = rng.normal(size=(n, 3))
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    tmax = np.where(rng.random(n) < 0.2, np.inf, rng.random(n) * 3.0 + 0.05).astype(np.float32)
    lam = (400 + 300 * rng.random((n, 4))).astype(np.float32)
    beta, ru, rl = [(0.3 + rng.random((n, 4))).astype(np.float32) for _ in range(3)]
    duc, du, iu = rng.random(n).astype(np.float32), rng.random((n, 2)).astype(np.float32), rng.random((n, 2)).astype(np.float32)
    osc = oracle.OracleScene(s)
    ref = osc.media_stage(0, depth, max_depth, np.concatenate([o, d, tmax[:, None], lam, beta, ru, rl], 1), duc, du, iu)
    last = osc.media_stage(0, max_depth - 1, max_depth, np.concatenate([o, d, tmax[:, None], lam, beta, ru, rl], 1)[:50], duc[:50], du[:50], iu[:50])
    osc.close()
    assert (ref[:, 0] == 1).sum() > 30 and ref[:, 17].sum() > (100 if which == "homogeneous" else 60) and ref[:, 36].sum() > (100 if which == "homogeneous" else 60)      # (an unbounded ray through a homogeneous medium never escapes)
    if which != "homogeneous":
        assert (ref[:, 0] == 2).sum() > 10                          # unbounded rays that leave the grid's bounds escape
    assert last[:, 17].sum() == 0                                   # at the depth limit nothing continues (K6: new_depth >= max_depth)

    def close(a, b):
        a, b = np.asarray(a, np.float32), np.asarray(b, np.float32)
        return bool(np.all(np.abs(a.view(np.int32).astype(np.int64) - b.view(np.int32).astype(np.int64)) <= 8) or np.allclose(a, b, rtol=2e-6, atol=0))

    bad = 0
    for i in range(n):
        v = R.media_vertex(sc, 0, o[i], d[i], tmax[i], lam[i], beta[i], ru[i], rl[i], depth, max_depth, duc[i], du[i], iu[i])
        r = ref[i]
        fate = {"survive": 1.0 if np.isfinite(tmax[i]) else 2.0}.get(v["kind"], 0.0)
        if v["kind"] == "survive" and (R.is_black(v["beta"][None])[0] or R.is_black(v["r_u"][None])[0]):
            fate = 0.0
        ok = fate == r[0] and close(v["add"], r[13:17])
        if fate in (1.0, 2.0):
            ok = ok and close(v["beta"], r[1:5]) and close(v["r_u"], r[5:9]) and close(v["r_l"], r[9:13])
        ok = ok and (v["cont"] is not None) == (r[17] == 1.0) and (v["shadow"] is not None) == (r[36] == 1.0)
        if ok and v["cont"] is not None:
            ok = close(v["cont"][0], r[18:21]) and close(v["cont"][1], r[21:24]) and close(v["beta"], r[24:28]) and close(v["r_u"], r[28:32]) and close(v["cont"][2], r[32:36])
        if ok and v["shadow"] is not None:
            so, sd, st_, sLd, sru, srl = v["shadow"]
            ok = close(so, r[37:40]) and close(sd, r[40:43]) and close([st_], r[43:44]) and close(sLd, r[44:48]) and close(sru, r[48:52]) and close(srl, r[52:56])
        bad += 0 if ok else 1
    print("media stage: %d rays, %d differ (fate, throughput, weights, continuation or shadow ray beyond 8 ulp)" % (n, bad))
    assert bad <= n // 100                                            # (a log / exp rounding that moves a sample across the segment end)


@pytest.mark.parametrize("variant,depth,spp", [("bake", 7, 4), ("analytic", 6, 4), ("no_sun", 5, 4)])
def test_sky_scene_per_pixel_against_the_numpy_restatement(hk, oracle, variant, depth, spp):
    """BASELINE configs[2] — the README scene: glass sphere, Gold(roughness 0.01) slab, Hosek-Wilkie sun-sky — pixel by pixel (round 5).
    The EnvironmentLight restated from textures/environment_map.jl:78-229, 290-371 (Clarberg's equal-area mapping with its polynomial
    atan, both directions; bilinear lookup by direction for an escaped ray, NEAREST lookup by uv for a sampled one), sampler/sampling.jl:
    207-361 (Distribution2D: marginal then conditional, the unrolled bisection, the discrete pdf) and physical-wavefront/lights.jl:158-190,
    336-347, 408-467 (pdf_image / 4 pi; an escaped ray's MIS weight with the light-choice pdf 1 / n_lights — Q6 — and the map's pdf);
    the SunLight (a delta infinite light with a baked RGBIlluminantSpectrum); both lights infinite in the sampler (p_inf = 2 / 2: no tree).
    Together with the glass sphere and the regularised gold lobe of the tests above: the 48 x 48 bake (64^2 map), the analytic map with
    a hot spot (importance sampling matters), and the map alone."""
    from hikari_jl_amd import scenes
    w = h = 32
    s, film, cam = scenes.sky_scene(w, h, env_res=64 if variant == "bake" else 48, tess=24, analytic=(variant == "analytic"), sun=(variant != "no_sun"))
    ref, img = _both(hk, oracle, s, cam, w, h, spp, depth)
    assert np.isfinite(img).all() and ref.max() > 0
    rel = np.sqrt(((img - ref) ** 2).sum(axis=2)) / (np.sqrt((ref ** 2).sum(axis=2)) + 1e-6)
    print("sky %s: pixels within 2e-4: %.4f, within 1e-2: %.4f, worst %.3g, mean ratio %.6f" % (variant, (rel <= 2e-4).mean(), (rel <= 1e-2).mean(), rel.max(), img.mean() / ref.mean()))
    assert (rel <= 2e-4).mean() >= 0.98 and (rel <= 1e-2).mean() >= 0.99
    assert abs(img.mean() / ref.mean() - 1.0) < 5e-3


@pytest.mark.gpu
@pytest.mark.parametrize("light,objects,depth,spp", [("area", "sphere_box", 5, 4), ("all", "sphere_box", 5, 4), ("area", "two_spheres", 4, 2)])
def test_device_frame_per_pixel_against_the_numpy_restatement(hk, light, objects, depth, spp):
    """The same comparison for the HIP path: the device's Cornell frame (sphere + box, depth 5, 4 spp, box filter; under the area light,
    and under area + point + spot + directional + ambient light; and SURVEY 8(d)'s two tessellated spheres — 3 782 triangles, the BVH that
    mixes LDS-cached and global nodes in one traversal) against the NumPy restatement, pixel by pixel — no oracle in between."""
    from hikari_jl_amd import scenes
    w = h = 32
    s, film, cam = scenes.cornell_box(w, h, light=light, objects=objects)
    vp = hk.VolPath(max_depth=depth, samples=spp, filter=hk.BoxFilter())
    vp(s, film, cam)
    dev = film.framebuffer.copy()
    mcv = float(vp.params.max_component_value)
    vp.close()
    img, _, _ = R.render(s.desc, cam.record(), hk.tables.load(), w, h, spp, depth, max_component_value=mcv, sobol_spp=spp)
    rel = np.sqrt(((img - dev) ** 2).sum(axis=2)) / (np.sqrt((dev ** 2).sum(axis=2)) + 1e-6)
    print("device vs restatement: within 2e-4: %.4f, within 1e-2: %.4f, worst %.3g" % ((rel <= 2e-4).mean(), (rel <= 1e-2).mean(), rel.max()))
    assert (rel <= 2e-4).mean() >= 0.99 and (rel <= 1e-2).mean() >= 0.995
    assert abs(img.mean() / dev.mean() - 1.0) < 1e-3


@pytest.mark.gpu
@pytest.mark.parametrize("metal,roughness,depth,spp", [("Copper", 0.3, 5, 4), ("Gold", 0.04, 6, 4)])
def test_device_conductor_per_pixel_against_the_numpy_restatement(hk, metal, roughness, depth, spp):
    """The HIP path's frame of the Cornell box with a rough metal sphere against the NumPy restatement, pixel by pixel — no oracle in between."""
    from hikari_jl_amd import scenes
    w = h = 32
    s, film, cam = scenes.cornell_box(w, h, light="area", object_material=getattr(hk, metal)(roughness=roughness))
    vp = hk.VolPath(max_depth=depth, samples=spp, filter=hk.BoxFilter())
    vp(s, film, cam)
    dev = film.framebuffer.copy()
    mcv = float(vp.params.max_component_value)
    vp.close()
    img, _, _ = R.render(s.desc, cam.record(), hk.tables.load(), w, h, spp, depth, max_component_value=mcv, sobol_spp=spp)
    rel = np.sqrt(((img - dev) ** 2).sum(axis=2)) / (np.sqrt((dev ** 2).sum(axis=2)) + 1e-6)
    print("device vs restatement (%s %.2f): within 2e-4: %.4f, within 1e-2: %.4f, worst %.3g" % (metal, roughness, (rel <= 2e-4).mean(), (rel <= 1e-2).mean(), rel.max()))
    assert (rel <= 2e-4).mean() >= 0.98 and (rel <= 1e-2).mean() >= 0.99
    assert abs(img.mean() / dev.mean() - 1.0) < 5e-3


@pytest.mark.gpu
@pytest.mark.parametrize("variant", ["bake", "analytic"])
def test_device_sky_scene_per_pixel_against_the_numpy_restatement(hk, variant):
    """The HIP path's frame of BASELINE configs[2] (glass sphere + gold slab + environment map + sun) against the NumPy restatement, pixel
    by pixel — no oracle in between."""
    from hikari_jl_amd import scenes
    w = h = 32
    s, film, cam = scenes.sky_scene(w, h, env_res=64 if variant == "bake" else 48, tess=24, analytic=(variant == "analytic"))
    vp = hk.VolPath(max_depth=6, samples=4, filter=hk.BoxFilter())
    vp(s, film, cam)
    dev = film.framebuffer.copy()
    mcv = float(vp.params.max_component_value)
    vp.close()
    img, _, _ = R.render(s.desc, cam.record(), hk.tables.load(), w, h, 4, 6, max_component_value=mcv, sobol_spp=4)
    rel = np.sqrt(((img - dev) ** 2).sum(axis=2)) / (np.sqrt((dev ** 2).sum(axis=2)) + 1e-6)
    print("device vs restatement (sky %s): within 2e-4: %.4f, within 1e-2: %.4f, worst %.3g" % (variant, (rel <= 2e-4).mean(), (rel <= 1e-2).mean(), rel.max()))
    assert (rel <= 2e-4).mean() >= 0.97 and (rel <= 1e-2).mean() >= 0.99
    assert abs(img.mean() / dev.mean() - 1.0) < 5e-3


def _many_light_small(scenes, w, h):
    """BASELINE configs[4] in small: the same barrel of boxes (matte, rough RGB-eta/k conductors, 25 % of them emissive with a per-face Le from
    a texture), 600 boxes scaled up to fill the frame: 7 200 triangles, 1 620 one-sided DiffuseAreaLights in the light BVH"""
    return scenes.many_light_scene(w, h, n_boxes=600, emissive_frac=0.25, box_scale=8.0)


def _leaf_depths(nodes16):
    leaf = nodes16[:, 14] > 0
    child = np.where(leaf, 0, nodes16[:, 13] - 1).astype(np.int64)
    depth = np.zeros(len(nodes16), np.int64)
    for k in range(len(nodes16)):
        if not leaf[k]:
            depth[k + 1] = depth[k] + 1
            depth[child[k]] = depth[k] + 1
    return depth[leaf]


def test_many_light_tree_and_frame_against_the_numpy_restatement(hk, oracle):
    """The many-light path from the second source, in two steps, because the reference's BUILD is ill-conditioned for this kind of scene.
    (1) The trees.  _evaluate_cost prices both sides of a split with the PARENT's area (bvh-light-sampler.jl:256-258), so splits differ only
    through the cones; boxes emit from opposite faces, the union of two opposite cones turns its axis about a x b (light-bounds.jl:79-87) and
    for a = -b that cross product is rounding noise — a few subtrees come out differently from two builders that both follow the text (here:
    libm against NumPy's float32 kernels rounded through binary64).  Measured: the same node count, 95 % of the bit trails identical, the
    same light for 98 % of random (point, normal, u).  The trees run deep (the cost ties send one bucket left and eleven right): 52 levels,
    a fifth of the leaves below level 32, where UInt32(1) << depth is 0 and the bit trail loses its turns (bvh-light-sampler.jl:457) — what
    bvh_pmf then walks is another leaf's path, in the reference as here.
    (2) The frame with the SAME tree under the restatement's own walks (bvh_sample_light, bvh_pmf by trail, importance): every lit pixel to
    rounding — 1 620 lights, the textured Le per face, MIS of emission that is hit against the pmf by (lossy) trail, conductors from RGB
    eta / k."""
    from hikari_jl_amd import scenes
    w = h = 24
    spp, depth = 2, 4
    s, film, cam = _many_light_small(scenes, w, h)
    p = hk.integrator_params(max_depth=depth, samples=spp, filter=hk.BoxFilter())
    osc = oracle.OracleScene(s)
    acc, _ = osc.render(p, cam, w, h, spp)
    ref = oracle.finalize(acc, w, h)
    nodes, trails = osc.light_bvh_nodes()
    sc = R.SceneNP(s.desc, R.Tables(hk.tables.load()))
    b = sc.bvh
    assert len(b.nodes) == len(nodes) == 2 * s.desc.n_lights - 1
    own = np.array([b.trail[i + 1] for i in range(s.desc.n_lights)], np.uint32)
    rng = np.random.default_rng(5)
    q = ((rng.random((4000, 3)) * 2 - 1) * 6).astype(np.float32)
    nn = rng.normal(size=(4000, 3))
    nn = (nn / np.linalg.norm(nn, axis=1, keepdims=True)).astype(np.float32)
    u = rng.random(4000).astype(np.float32)
    l_np, pmf_np = b.sample(q, nn, u)
    l_or, pmf_or, _ = osc.light_bvh(q, nn, u)
    osc.close()
    deep = _leaf_depths(nodes)
    print("trails equal %.4f, same light %.4f, max depth %d, leaves below level 32: %d of %d" % ((own == trails).mean(), (l_np == l_or).mean(), deep.max(), (deep > 32).sum(), len(deep)))
    assert (own == trails).mean() >= 0.9 and (l_np == l_or).mean() >= 0.95
    assert deep.max() > 32 and deep.max() < 64                                    # lossy trails are exercised; the device's walks stop at 64 levels
    b.adopt(nodes, trails)
    l_ad, pmf_ad = b.sample(q, nn, u)
    hit = pmf_or > 0
    assert (l_ad == l_or).mean() >= 0.999 and np.abs(pmf_ad[hit & (l_ad == l_or)] / pmf_or[hit & (l_ad == l_or)] - 1).max() < 1e-3
    img, _, _ = R.render(s.desc, cam.record(), hk.tables.load(), w, h, spp, depth, max_component_value=float(p.max_component_value), sobol_spp=spp, scene=sc)
    lit = ref.sum(axis=2) > 0
    rel = np.sqrt(((img - ref) ** 2).sum(axis=2)) / (np.sqrt((ref ** 2).sum(axis=2)) + 1e-6)
    print("lit pixels %d of %d; within 2e-4: %.4f, worst %.3g, mean ratio %.6f" % (lit.sum(), lit.size, (rel[lit] <= 2e-4).mean(), rel.max(), img.mean() / ref.mean()))
    # measured: 321 lit pixels, all within 1.5e-5
    assert lit.sum() >= 250 and np.array_equal(lit, img.sum(axis=2) > 0)
    assert (rel[lit] <= 2e-4).mean() >= 0.99 and (rel[lit] <= 1e-2).mean() >= 0.995
    assert abs(img.mean() / ref.mean() - 1.0) < 1e-3


@pytest.mark.gpu
def test_device_many_light_frame_against_the_numpy_restatement(hk, gpu_ctx):
    """The HIP path's frame of the small many-light scene against the NumPy restatement walking the LIBRARY's own light BVH (read back through
    hk_scene_light_bvh_copy; the build is compared with the restatement's in the CPU test above) — no oracle in between."""
    import ctypes as C
    from hikari_jl_amd import scenes
    w = h = 24
    s, film, cam = _many_light_small(scenes, w, h)
    vp = hk.VolPath(max_depth=4, samples=2, filter=hk.BoxFilter())
    vp(s, film, cam)
    dev = film.framebuffer.copy()
    mcv = float(vp.params.max_component_value)
    vp.close()
    sh = hk.scene_handle(gpu_ctx, s)
    L = hk._lib.lib()
    nn = C.c_int32()
    L.hk_scene_light_bvh_copy(sh, C.byref(nn), None, None)
    nodes = np.zeros((nn.value, 16), np.float32)
    trails = np.zeros(s.desc.n_lights, np.uint32)
    L.hk_scene_light_bvh_copy(sh, C.byref(nn), nodes.ctypes.data_as(hk._abi.PF), trails.ctypes.data_as(C.POINTER(C.c_uint32)))
    sc = R.SceneNP(s.desc, R.Tables(hk.tables.load()))
    sc.bvh.adopt(nodes, trails)
    img, _, _ = R.render(s.desc, cam.record(), hk.tables.load(), w, h, 2, 4, max_component_value=mcv, sobol_spp=2, scene=sc)
    lit = dev.sum(axis=2) > 0
    rel = np.sqrt(((img - dev) ** 2).sum(axis=2)) / (np.sqrt((dev ** 2).sum(axis=2)) + 1e-6)
    print("device vs restatement (many-light): lit %d, within 2e-4: %.4f, worst %.3g" % (lit.sum(), (rel[lit] <= 2e-4).mean(), rel.max()))
    assert lit.sum() >= 250
    assert (rel[lit] <= 2e-4).mean() >= 0.99 and (rel[lit] <= 1e-2).mean() >= 0.995
    assert abs(img.mean() / dev.mean() - 1.0) < 1e-3


def _textured_room(hk, w, h, full=False):
    """a room whose floor, back wall and sphere carry image textures of different, non-square sizes (a transposed or flipped lookup shows),
    under an area light and a point light"""
    from hikari_jl_amd import geometry as G
    from hikari_jl_amd.materials import Texture
    R = hk.RGBSpectrum
    rng = np.random.default_rng(12)
    s = hk.Scene()
    smooth = lambda hh, ww: (0.15 + 0.8 * rng.random((hh, ww, 3))).astype(np.float32)
    s.push(G.quad((-1, 0, -1), (1, 0, -1), (1, 0, 1), (-1, 0, 1), normal=(0, 1, 0)), hk.MatteMaterial(Kd=Texture(smooth(6, 5))))
    s.push(G.quad((-1, 0, -1), (-1, 2, -1), (1, 2, -1), (1, 0, -1), normal=(0, 0, 1)), hk.MatteMaterial(Kd=Texture(smooth(3, 7))))
    s.push(G.quad((-1, 0, 1), (-1, 2, 1), (-1, 2, -1), (-1, 0, -1), normal=(1, 0, 0)), hk.MatteMaterial(Kd=R(0.7, 0.2, 0.2)))
    s.push(G.quad((1, 0, -1), (1, 2, -1), (1, 2, 1), (1, 0, 1), normal=(-1, 0, 0)), hk.MatteMaterial(Kd=R(0.2, 0.6, 0.25)))
    s.push(G.quad((-1, 2, -1), (-1, 2, 1), (1, 2, 1), (1, 2, -1), normal=(0, -1, 0)), hk.MatteMaterial(Kd=R(0.75)))
    s.push(G.sphere((0.25, 0.45, 0.1), 0.45, 12), hk.MatteMaterial(Kd=Texture(smooth(8, 4))))
    if full:      # every textured parameter the restatement reads: sigma (a float image), vertex colours, Kr, Kt, a conductor's roughness
        from hikari_jl_amd.materials import VertexColorTexture
        sig = (30.0 * rng.random((4, 3))).astype(np.float32)
        s.push(G.quad((-0.95, 0.02, 0.2), (-0.4, 0.02, 0.2), (-0.4, 0.02, 0.9), (-0.95, 0.02, 0.9), normal=(0, 1, 0)), hk.MatteMaterial(Kd=Texture(smooth(3, 3)), sigma=Texture(sig)))
        panel = G.rect3f((-0.9, 0.9, -0.95), (0.6, 0.7, 0.05))
        s.push(panel, hk.MatteMaterial(Kd=VertexColorTexture((0.2 + 0.7 * rng.random((panel.n_faces, 3, 3))).astype(np.float32))))
        s.push(G.quad((0.55, 0.6, -0.9), (0.95, 0.6, -0.6), (0.95, 1.5, -0.6), (0.55, 1.5, -0.9)), hk.MirrorMaterial(Kr=Texture(smooth(4, 4))))
        s.push(G.sphere((-0.45, 0.3, -0.3), 0.3, 10), hk.GlassMaterial(Kr=Texture(smooth(2, 3)), Kt=Texture(smooth(3, 2)), index=1.5))
        s.push(G.sphere((0.65, 0.25, 0.6), 0.25, 10), hk.ConductorMaterial(eta=R(0.2, 0.92, 1.1), k=R(3.9, 2.45, 2.14), roughness=Texture((0.05 + 0.5 * rng.random((5, 4))).astype(np.float32))))
    s.push(G.quad((-0.3, 1.98, -0.3), (0.3, 1.98, -0.3), (0.3, 1.98, 0.3), (-0.3, 1.98, 0.3), normal=(0, -1, 0)),
           hk.MediumInterface(hk.MatteMaterial(Kd=R(0.0)), emission=hk.Emissive(Le=R(1.0, 0.9, 0.8), scale=12.0)))
    s.push(hk.PointLight((-0.6, 1.2, 0.6), R(2.0, 2.5, 3.0)))
    s.sync()
    film = hk.Film((w, h))
    cam = hk.PerspectiveCamera((0.0, 1.0, 3.6), (0.0, 0.9, 0.0), film, up=(0, 1, 0), fov=40.0)
    return s, film, cam


@pytest.mark.parametrize("full", [False, True])
def test_textured_matte_per_pixel_against_the_numpy_restatement(hk, oracle, full):
    """TEXTURES inside the loop: a Matte's Kd from an image at the hit's uv — the barycentric uv (physical-wavefront/intersection.jl:181-194),
    the bilinear lookup with its (1 - v, u) flip and clamped neighbours (textures/texture-ref.jl:151-186), Kd clamped and uplifted per
    vertex (spectral-eval.jl:57-63) — for next-event estimation and for the sampled bounce alike."""
    w = h = 32
    s, film, cam = _textured_room(hk, w, h, full)     # full: + a sigma image (the SAMPLED lobe scaled, the evaluated one not: spectral-eval.jl:88-96 against
    ref, img = _both(hk, oracle, s, cam, w, h, 4, 5)  # :372-396), a VertexColorTexture (texture-ref.jl:230-235), textured Kr / Kt / conductor roughness
    assert np.isfinite(img).all() and ref.max() > 0
    rel = np.sqrt(((img - ref) ** 2).sum(axis=2)) / (np.sqrt((ref ** 2).sum(axis=2)) + 1e-6)
    print("pixels within 2e-4: %.4f, within 1e-2: %.4f, worst %.3g, mean ratio %.6f" % ((rel <= 2e-4).mean(), (rel <= 1e-2).mean(), rel.max(), img.mean() / ref.mean()))
    assert (rel <= 2e-4).mean() >= 0.99 and (rel <= 1e-2).mean() >= 0.995
    assert abs(img.mean() / ref.mean() - 1.0) < 1e-3


@pytest.mark.gpu
def test_device_textured_matte_per_pixel_against_the_numpy_restatement(hk):
    """the HIP path's frame of the textured room against the NumPy restatement — no oracle in between"""
    w = h = 32
    s, film, cam = _textured_room(hk, w, h, True)
    vp = hk.VolPath(max_depth=5, samples=4, filter=hk.BoxFilter())
    vp(s, film, cam)
    dev = film.framebuffer.copy()
    mcv = float(vp.params.max_component_value)
    vp.close()
    img, _, _ = R.render(s.desc, cam.record(), hk.tables.load(), w, h, 4, 5, max_component_value=mcv, sobol_spp=4)
    rel = np.sqrt(((img - dev) ** 2).sum(axis=2)) / (np.sqrt((dev ** 2).sum(axis=2)) + 1e-6)
    print("device vs restatement (textured room): within 2e-4: %.4f, within 1e-2: %.4f, worst %.3g" % ((rel <= 2e-4).mean(), (rel <= 1e-2).mean(), rel.max()))
    assert (rel <= 2e-4).mean() >= 0.99 and (rel <= 1e-2).mean() >= 0.995
    assert abs(img.mean() / dev.mean() - 1.0) < 1e-3


@pytest.mark.gpu
@pytest.mark.parametrize("which", ["grid", "grid_rotated", "nanovdb", "rgbgrid"])
def test_device_media_points_and_segments_against_the_numpy_restatement(hk, gpu_ctx, which):
    """The HIP path's heterogeneous media against the NumPy restatement directly (no oracle in between): sample_point — the trilinear
    density of a GridMedium, the NanoVDB tree decoded from its bytes and its index-space sampler, the unbounded uplift of sigma_a / sigma_s —
    at 3 000 points, and the majorant segments a ray meets (ray to medium space, ray / bounds, the DDA over the majorant cells with the
    grid the restatement REBUILT from the density / the tree) for 600 rays: segment count, and t_min / t_max / sigma_maj of the first 16."""
    import ctypes as C
    from hikari_jl_amd import scenes
    gr = np.random.default_rng(8)
    bounds = ((-2.5, -2.6, 1.0), (2.5, 2.6, 2.0))
    kw = dict(sigma_a=hk.RGBSpectrum(0.3, 0.4, 0.2), sigma_s=hk.RGBSpectrum(1.2, 1.0, 1.4), g=0.35, majorant_res=(4, 3, 5))
    if which == "rgbgrid":
        ga = (gr.random((7, 6, 5, 3)) * 0.5).astype(np.float32)
        gs = (gr.random((7, 6, 5, 3)) * 1.5).astype(np.float32)
        gl = (gr.random((7, 6, 5, 3)) * 0.2).astype(np.float32)
        gs[:2, :, :2] = 0.0
        ga[:2, :, :2] = 0.0
        med = hk.RGBGridMedium(sigma_a_grid=ga, sigma_s_grid=gs, Le_grid=gl, sigma_scale=1.7, Le_scale=0.8, g=-0.2, bounds=bounds, majorant_res=(3, 4, 2))
    elif which == "nanovdb":
        big = (gr.random((24, 17, 11)) ** 2 * 2.5).astype(np.float32)
        big[:7, :, :4] = 0.0
        med = hk.NanoVDBMedium(big, bounds=bounds, **kw)
    else:
        dens = (gr.random((10, 8, 6)) ** 2 * 2.5).astype(np.float32)
        dens[:3, :, :2] = 0.0
        xf = None
        if which == "grid_rotated":
            c, sn = np.cos(0.3), np.sin(0.3)
            xf = np.array([[c, -sn, 0, 0.2], [sn, c, 0, -0.1], [0, 0, 1, 0.05], [0, 0, 0, 1]], np.float32)
        med = hk.GridMedium(dens, bounds=bounds, transform=xf, **kw)
    s, _, _ = scenes.slab_scene(16, 16, med)
    md = R.SceneNP(s.desc, R.Tables(hk.tables.load())).media[0]
    sh = hk.scene_handle(gpu_ctx, s)
    L = hk._lib.lib()
    PF = hk._abi.PF
    pf = lambda a: a.ctypes.data_as(PF)
    rng = np.random.default_rng(21)
    n = 3000
    p = (rng.random((n, 3)) * np.array([5.6, 5.8, 1.3]) + np.array([-2.8, -2.9, 0.85])).astype(np.float32)
    lam = (380 + 420 * rng.random((n, 4))).astype(np.float32)
    out = np.zeros((n, 13), np.float32)
    hk._lib.check(L.hk_test_medium(gpu_ctx.h, sh, 0, 0, n, pf(p), None, None, pf(lam), pf(out)), "hk_test_medium")
    cols = 12 if which == "rgbgrid" else 8                                       # (the RGB grid medium emits: Le too)
    mine = np.array([np.concatenate(md.point(p[i], lam[i])[:cols // 4]) for i in range(n)]).astype(np.float32)
    ulp = np.abs(out[:, :cols].view(np.int32).astype(np.int64) - mine.view(np.int32).astype(np.int64))
    print("device vs restatement (%s): sample_point max %d ulp over %d points, %.2f of them inside the medium" % (which, ulp.max(), n, (mine[:, 0] > 0).mean()))
    assert (mine[:, 0] > 0).mean() > 0.4 and ulp.max() <= (4 if which == "rgbgrid" else 2)       # (the device's per-point uplift runs the hardware exp)
    m = 600
    o = (rng.random((m, 3)) * np.array([7.0, 7.0, 3.0]) + np.array([-3.5, -3.5, 0.0])).astype(np.float32)
    tgt = (rng.random((m, 3)) * np.array([5.0, 5.2, 1.0]) + np.array([-2.5, -2.6, 1.0])).astype(np.float32)
    d = tgt - o
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    d[:40] = np.array([0, 0, 1], np.float32)
    tmax = np.where(rng.random(m) < 0.3, rng.random(m) * 3.0, np.inf).astype(np.float32)
    outm = np.zeros((m, 49), np.float32)
    hk._lib.check(L.hk_test_medium(gpu_ctx.h, sh, 1, 0, m, pf(o), pf(d), pf(tmax), pf(lam[:m]), pf(outm)), "hk_test_medium")
    bad = 0
    for i in range(m):
        segs = []
        for t0, t1, smaj in md.segments(o[i], d[i], tmax[i], lam[i]):
            segs.append((t0, t1, smaj[0]))
            if len(segs) >= 256:
                break
        ok = int(outm[i, 0]) == len(segs)
        got = outm[i, 1:].reshape(16, 3)[:min(len(segs), 16)]
        want = np.array(segs[:16], np.float32).reshape(-1, 3)
        if ok and len(want):
            ok = bool((np.abs(got.view(np.int32).astype(np.int64) - want.view(np.int32).astype(np.int64)) <= 2).all())
        bad += 0 if ok else 1
    print("device vs restatement (%s): majorant segments of %d rays, %d differ; up to %d segments per ray" % (which, m, bad, int(outm[:, 0].max())))
    assert outm[:, 0].max() >= 4 and bad == 0


def _thin_and_translucent_box(hk, scenes, w, h, light):
    """the Cornell box with a ThinDielectric pane and a DiffuseTransmission sheet standing in it (both two-sided in effect: light reaches
    the walls behind them by transmission), the sphere kept matte"""
    from hikari_jl_amd import geometry as G
    R = hk.RGBSpectrum
    s, film, cam = scenes.cornell_box(w, h, light=light, spheres=False)
    s.push(G.quad((-0.7, 0.0, 0.35), (-0.05, 0.0, 0.05), (-0.05, 1.3, 0.05), (-0.7, 1.3, 0.35)), hk.ThinDielectricMaterial(eta=1.45))
    s.push(G.quad((0.1, 0.0, 0.0), (0.75, 0.0, 0.3), (0.75, 1.1, 0.3), (0.1, 1.1, 0.0)),
           hk.DiffuseTransmissionMaterial(reflectance=R(0.5, 0.35, 0.2), transmittance=R(0.3, 0.5, 0.7), scale=1.3))
    s.push(G.sphere((0.0, 0.3, 0.55), 0.3, 10), hk.MatteMaterial(Kd=R(0.6)))
    s.sync()
    return s, film, cam


@pytest.mark.parametrize("light,depth,spp", [("area", 6, 4), ("both", 5, 4)])
def test_thin_dielectric_and_diffuse_transmission_per_pixel_against_the_numpy_restatement(hk, oracle, light, depth, spp):
    """Two more material kinds inside the loop: ThinDielectric (spectral-eval.jl:1975-2037: slab reflectance R0 + T0^2 R0 / (1 - R0^2), the
    lobe chosen by the bounce's 1-D sample, f = R / |cos| or T / |cos| with beta *= f as the reference has it) and DiffuseTransmission
    (:2083-2215: reflectance / transmittance times scale, clamped; the side chosen by the larger components; evaluated on BOTH sides in
    next-event estimation, with the lobe probability in the pdf that enters the MIS weights)."""
    from hikari_jl_amd import scenes
    w = h = 32
    s, film, cam = _thin_and_translucent_box(hk, scenes, w, h, light)
    ref, img = _both(hk, oracle, s, cam, w, h, spp, depth)
    assert np.isfinite(img).all() and ref.max() > 0
    rel = np.sqrt(((img - ref) ** 2).sum(axis=2)) / (np.sqrt((ref ** 2).sum(axis=2)) + 1e-6)
    print("pixels within 2e-4: %.4f, within 1e-2: %.4f, worst %.3g, mean ratio %.6f" % ((rel <= 2e-4).mean(), (rel <= 1e-2).mean(), rel.max(), img.mean() / ref.mean()))
    assert (rel <= 2e-4).mean() >= 0.99 and (rel <= 1e-2).mean() >= 0.995
    assert abs(img.mean() / ref.mean() - 1.0) < 1e-3


@pytest.mark.gpu
def test_device_thin_dielectric_and_diffuse_transmission_against_the_numpy_restatement(hk):
    """the HIP path's frame of that box against the NumPy restatement — no oracle in between"""
    from hikari_jl_amd import scenes
    w = h = 32
    s, film, cam = _thin_and_translucent_box(hk, scenes, w, h, "both")
    vp = hk.VolPath(max_depth=5, samples=4, filter=hk.BoxFilter())
    vp(s, film, cam)
    dev = film.framebuffer.copy()
    mcv = float(vp.params.max_component_value)
    vp.close()
    img, _, _ = R.render(s.desc, cam.record(), hk.tables.load(), w, h, 4, 5, max_component_value=mcv, sobol_spp=4)
    rel = np.sqrt(((img - dev) ** 2).sum(axis=2)) / (np.sqrt((dev ** 2).sum(axis=2)) + 1e-6)
    print("device vs restatement (thin dielectric + diffuse transmission): within 2e-4: %.4f, within 1e-2: %.4f, worst %.3g" % ((rel <= 2e-4).mean(), (rel <= 1e-2).mean(), rel.max()))
    assert (rel <= 2e-4).mean() >= 0.99 and (rel <= 1e-2).mean() >= 0.995
    assert abs(img.mean() / dev.mean() - 1.0) < 1e-3


@pytest.mark.parametrize("which", ["rough_rough", "smooth_rough_reflectance"])
def test_coated_conductor_per_pixel_against_the_numpy_restatement(hk, oracle, which):
    """A LAYERED kind inside the loop: the reference's CoatedConductor is an analytic composition of coating and base (spectral-eval.jl:
    2877-3412; no random walk, no hashed seeds), so its bounce can be followed per pixel — next-event estimation through cc_eval with the
    pdf that enters the MIS weights, the sampled bounce through cc_sample (r_l = r_u / pdf after it), regularised once the path has had
    a non-specular bounce.  (CoatedDiffuse and CoatedDiffuseTransmission ARE hashed walks: point-wise only, tests/test_layered_pin.py.)"""
    from hikari_jl_amd import scenes
    R_ = hk.RGBSpectrum
    w = h = 32
    if which == "rough_rough":
        mat = hk.CoatedConductorMaterial(interface_u_roughness=0.1, interface_v_roughness=0.1, interface_eta=1.5, conductor_u_roughness=0.2, conductor_v_roughness=0.2,
                                         albedo=R_(0.7, 0.8, 0.9), thickness=0.05)
    else:
        mat = hk.CoatedConductorMaterial(interface_eta=1.45, conductor_u_roughness=0.15, conductor_v_roughness=0.3, reflectance=R_(0.9, 0.6, 0.3))
    s, film, cam = scenes.cornell_box(w, h, light="both", object_material=mat)
    ref, img = _both(hk, oracle, s, cam, w, h, 4, 6)
    assert np.isfinite(img).all() and ref.max() > 0
    rel = np.sqrt(((img - ref) ** 2).sum(axis=2)) / (np.sqrt((ref ** 2).sum(axis=2)) + 1e-6)
    print("pixels within 2e-4: %.4f, within 1e-2: %.4f, worst %.3g, mean ratio %.6f" % ((rel <= 2e-4).mean(), (rel <= 1e-2).mean(), rel.max(), img.mean() / ref.mean()))
    assert (rel <= 2e-4).mean() >= 0.99 and (rel <= 1e-2).mean() >= 0.995
    assert abs(img.mean() / ref.mean() - 1.0) < 1e-3


@pytest.mark.gpu
def test_device_coated_conductor_per_pixel_against_the_numpy_restatement(hk):
    """the HIP path's frame of the box with a rough-on-rough CoatedConductor object against the NumPy restatement — no oracle in between"""
    from hikari_jl_amd import scenes
    R_ = hk.RGBSpectrum
    w = h = 32
    mat = hk.CoatedConductorMaterial(interface_u_roughness=0.1, interface_v_roughness=0.1, interface_eta=1.5, conductor_u_roughness=0.2, conductor_v_roughness=0.2,
                                     albedo=R_(0.7, 0.8, 0.9), thickness=0.05)
    s, film, cam = scenes.cornell_box(w, h, light="both", object_material=mat)
    vp = hk.VolPath(max_depth=6, samples=4, filter=hk.BoxFilter())
    vp(s, film, cam)
    dev = film.framebuffer.copy()
    mcv = float(vp.params.max_component_value)
    vp.close()
    img, _, _ = R.render(s.desc, cam.record(), hk.tables.load(), w, h, 4, 6, max_component_value=mcv, sobol_spp=4)
    rel = np.sqrt(((img - dev) ** 2).sum(axis=2)) / (np.sqrt((dev ** 2).sum(axis=2)) + 1e-6)
    print("device vs restatement (coated conductor in the box): within 2e-4: %.4f, within 1e-2: %.4f, worst %.3g" % ((rel <= 2e-4).mean(), (rel <= 1e-2).mean(), rel.max()))
    assert (rel <= 2e-4).mean() >= 0.99 and (rel <= 1e-2).mean() >= 0.995
    assert abs(img.mean() / dev.mean() - 1.0) < 1e-3


CD_CASES = {
    "cd_rough": lambda hk: hk.CoatedDiffuseMaterial(reflectance=hk.RGBSpectrum(0.6, 0.3, 0.2), u_roughness=0.1, v_roughness=0.1, thickness=0.01, eta=1.5),
    "cd_smooth_medium": lambda hk: hk.CoatedDiffuseMaterial(reflectance=hk.RGBSpectrum(0.4, 0.5, 0.6), thickness=0.05, eta=1.33, albedo=hk.RGBSpectrum(0.6, 0.7, 0.8), g=0.3, n_samples=2),
    "cdt": lambda hk: hk.CoatedDiffuseTransmissionMaterial(reflectance=hk.RGBSpectrum(0.5, 0.3, 0.2), transmittance=hk.RGBSpectrum(0.2, 0.3, 0.4), u_roughness=0.15, v_roughness=0.15,
                                                            thickness=0.01, eta=1.5),
}


@pytest.mark.parametrize("which", list(CD_CASES))
def test_coated_diffuse_inside_the_loop_against_the_numpy_restatement(hk, oracle, which):
    """Round 6 (the CoatedDiffuse / CoatedDiffuseTransmission walks were single-sourced INSIDE the loop): the LayeredBxDF
    random walks of ref_layered_np now run inside the NumPy wavefront loop — evaluate (the nSamples walks with their NEE and MIS, its pdf
    estimate entering the path's MIS weights) at every next-event estimation, sample (the walk that picks the continuation, its pdf in
    r_l = r_u / pdf) at every bounce, regularised once the path has had a non-specular bounce.  The walks seed a PCG32 from the float BITS
    of wo / wi in the SHADING FRAME and of the samples (spectral-eval.jl:1316, 1636): one unit in the last place anywhere upstream and the
    two sides walk differently.  The restatement therefore intersects in binary32 here (SceneNP.intersect32) and — since this round —
    normalises and applies the homogeneous divide as `inv(norm) * v` / one reciprocal (StaticArrays' normalize; it divided until the hashed
    decisions exposed the last-bit difference on a tenth of the camera rays): camera rays are bit-equal to the oracle's now.
      * depth 2, 4 spp (the camera vertex's next-event estimation and walk, the next vertex's next-event estimation): >= 95 % of the pixels
        within 2e-4 (measured 96.9 - 98.6 %; depth 1: all of them), frame means within 0.1 %;
      * depth 6, 64 spp: per-channel frame means within 1 % (measured 0.1 - 0.5 %) — a wrong pdf in the MIS weights, a missing regularisation
        or r_l not divided by the walk's pdf moves them by several per cent."""
    from hikari_jl_amd import scenes
    w = h = 24
    s, film, cam = scenes.cornell_box(w, h, light="both", object_material=CD_CASES[which](hk))
    ref, img = _both(hk, oracle, s, cam, w, h, 4, 2, hits32=True)
    assert np.isfinite(img).all() and ref.max() > 0
    rel = np.sqrt(((img - ref) ** 2).sum(axis=2)) / (np.sqrt((ref ** 2).sum(axis=2)) + 1e-6)
    print("%s depth 2: pixels within 2e-4: %.4f, within 1e-2: %.4f, mean ratio %.6f" % (which, (rel <= 2e-4).mean(), (rel <= 1e-2).mean(), img.mean() / ref.mean()))
    assert (rel <= 2e-4).mean() >= 0.95 and abs(img.mean() / ref.mean() - 1.0) < 1e-3
    ref, img = _both(hk, oracle, s, cam, 12, 12, 64, 6, hits32=True)
    ratio = img.mean(axis=(0, 1)) / ref.mean(axis=(0, 1))
    print("%s depth 6, 64 spp: channel mean ratios %s" % (which, ratio.round(5)))
    assert np.isfinite(img).all() and (np.abs(ratio - 1.0) < 0.01).all()


@pytest.mark.parametrize("amount", [0.35, 0.7])
def test_mix_material_inside_the_loop_against_the_numpy_restatement(hk, oracle, amount):
    """Round 6: MixMaterial inside the NumPy wavefront loop.  The sphere of the box is Mix(matte, mirror) nested in Mix(., glass): each hit hashes
    the bits of the hit point, of wo and of the children's keys (tests/test_hash_pins.py::mix_hash_float, integers only) against the amount
    and shades the chosen child — a wrong point (the ray origin instead of the hit), a wrong wo sign or a swapped comparison re-deals every
    choice.  The restatement intersects in binary32 here (SceneNP.intersect32) so that the hashed bits are the oracle's wherever the path
    up to the hit is: the camera vertex always (depth 1: every pixel agrees); deeper vertices while no rounding differed (depth 5, 4 spp: 94 - 96 %
    of the pixels within 2e-4, frame means within 0.1 %)."""
    from hikari_jl_amd import scenes
    R_ = hk.RGBSpectrum
    inner = hk.MixMaterial((hk.MatteMaterial(Kd=R_(0.7, 0.3, 0.2)), hk.MirrorMaterial(Kr=R_(0.9, 0.85, 0.7))), amount)
    outer = hk.MixMaterial((inner, hk.GlassMaterial(Kr=R_(1.0), Kt=R_(1.0), index=1.5)), 0.25)
    w = h = 32
    s, film, cam = scenes.cornell_box(w, h, light="both", object_material=outer)
    ref, img = _both(hk, oracle, s, cam, w, h, 4, 5, hits32=True)
    assert np.isfinite(img).all() and ref.max() > 0
    rel = np.sqrt(((img - ref) ** 2).sum(axis=2)) / (np.sqrt((ref ** 2).sum(axis=2)) + 1e-6)
    print("mix %.2f: pixels within 2e-4: %.4f, within 1e-2: %.4f, mean ratio %.6f" % (amount, (rel <= 2e-4).mean(), (rel <= 1e-2).mean(), img.mean() / ref.mean()))
    assert (rel <= 2e-4).mean() >= 0.92 and abs(img.mean() / ref.mean() - 1.0) < 5e-3

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


def _both(hk, oracle, s, cam, w, h, spp, depth):
    p = hk.integrator_params(max_depth=depth, samples=spp, filter=hk.BoxFilter())
    osc = oracle.OracleScene(s)
    acc, _ = osc.render(p, cam, w, h, spp)
    osc.close()
    ref = oracle.finalize(acc, w, h)
    img, _, _ = R.render(s.desc, cam.record(), hk.tables.load(), w, h, spp, depth, max_component_value=float(p.max_component_value), sobol_spp=spp)
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

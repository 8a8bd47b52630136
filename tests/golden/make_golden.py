#!/usr/bin/env python3
"""Regenerates tests/golden/*.npz.  Run from the repository root:  python tests/golden/make_golden.py

The reference (Julia + Raycore) cannot run in this image, so these vectors are produced by the build's OWN CPU oracle
(oracle/, a restatement of the reference's algorithm — "parity unpinned") and are REGRESSION fixtures:
they pin today's oracle so that neither it nor the HIP path can drift unnoticed, they are not reference truth.
What IS reference data: data_tables.json holds the SHA-256 of the numeric tables extracted verbatim from the reference source
(tools/extract_reference_tables.py: Sobol matrices, CIE XYZ, metal spectra, Hosek-Wilkie coefficients)."""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)
HERE = os.path.dirname(os.path.abspath(__file__))

# name -> (scene builder, integrator keywords, (w, h))
def cases():
    import hikari_jl_amd as hk
    from hikari_jl_amd import scenes
    def slab(w, h):
        return scenes.slab_scene(w, h, hk.HomogeneousMedium(sigma_a=hk.RGBSpectrum(0.2), sigma_s=hk.RGBSpectrum(0.8, 0.7, 0.6), g=0.3))
    return {
        "single_triangle": (lambda w, h: scenes.single_triangle(w, h), dict(max_depth=4, samples=4), (40, 30)),
        "cornell_area": (lambda w, h: scenes.cornell_box(w, h, light="area"), dict(max_depth=5, samples=4), (32, 32)),
        "cornell_point": (lambda w, h: scenes.cornell_box(w, h, light="point"), dict(max_depth=5, samples=4), (32, 32)),
        "coated_diffuse": (lambda w, h: scenes.material_scene(w, h, hk.CoatedDiffuseMaterial(reflectance=hk.RGBSpectrum(0.2, 0.5, 0.7), u_roughness=0.1, v_roughness=0.1)),
                           dict(max_depth=6, samples=2), (36, 24)),
        "glass": (lambda w, h: scenes.material_scene(w, h, hk.GlassMaterial(index=1.5)), dict(max_depth=8, samples=2), (36, 24)),
        "sky": (lambda w, h: scenes.sky_scene(w, h, env_res=32), dict(max_depth=6, samples=2), (32, 32)),
        "slab_homogeneous": (slab, dict(max_depth=6, samples=4), (24, 24)),
    }


STATISTICAL = ("slab_homogeneous", "coated_diffuse")   # RNG streams seeded from float bit patterns: compared through an independent second frame


def render(name, first=1):
    import hikari_jl_amd as hk
    import oracle
    build, kw, (w, h) = cases()[name]
    scene, film, cam = build(w, h)
    osc = oracle.OracleScene(scene)
    acc, st = osc.render(hk.integrator_params(**kw), cam, w, h, kw["samples"], first=first)
    osc.close()
    return oracle.finalize(acc, w, h), np.array([int(st.rays_closest), int(st.rays_shadow), int(st.path_vertices)], np.int64)


def kat_vectors():
    """sub-kernel known answers of the oracle on fixed inputs (Sobol draws, RGB uplift)"""
    import hikari_jl_amd as hk
    import oracle
    rng = np.random.default_rng(2024)
    n = 64
    px, py = rng.integers(1, 800, n).astype(np.int32), rng.integers(1, 800, n).astype(np.int32)
    idx, dim = rng.integers(1, 256, n).astype(np.int32), rng.integers(0, 60, n).astype(np.int32)
    s1, s2 = oracle.sobol(800, 800, 256, 0, px, py, idx, dim)
    rgb = rng.random((n, 3)).astype(np.float32)
    lam = (360.0 + 470.0 * rng.random((n, 4))).astype(np.float32)
    up = [oracle.uplift(mode, rgb, lam) for mode in (0, 1, 2)]
    return dict(px=px, py=py, idx=idx, dim=dim, sobol_1d=s1, sobol_2d=s2, rgb=rgb, lam=lam, uplift_bounded=up[0], uplift_unbounded=up[1], uplift_illuminant=up[2])


def table_hashes():
    d = os.path.join(ROOT, "hikari.jl_amd", "data")
    return {f: hashlib.sha256(open(os.path.join(d, f), "rb").read()).hexdigest() for f in ("sobol_matrices.bin", "cie_xyz.bin", "metal_spectra.bin", "hosek_wilkie_sky.bin", "medium_presets.json")}


def main():
    import oracle
    oracle.build()
    for name in cases():
        img, counts = render(name)
        extra = {}
        if name in STATISTICAL:   # frame B: the NEXT `samples` sample indices — the yardstick of the statistical comparison
            extra["framebuffer_b"] = render(name, first=cases()[name][1]["samples"] + 1)[0]
        np.savez_compressed(os.path.join(HERE, name + ".npz"), framebuffer=img, counts=counts, **extra)
        print(name, img.shape, float(img.mean()), counts)
    np.savez_compressed(os.path.join(HERE, "kat_vectors.npz"), **kat_vectors())
    json.dump(table_hashes(), open(os.path.join(HERE, "data_tables.json"), "w"), indent=1)


if __name__ == "__main__":
    main()

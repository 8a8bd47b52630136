#!/usr/bin/env python3
"""Inputs of the reference-fixture generator (julia/make_reference_fixtures.jl): fixed pseudo-random points of every per-stage function
the oracle restates, written to tests/golden/reference_inputs/ in the raw format of tests/fixture_io.py.  Deterministic (seeded): the
committed files are exactly what this script writes (tests/test_reference_fixtures.py::test_inputs_are_what_the_script_writes).

    python tests/golden/make_reference_inputs.py            # rewrites tests/golden/reference_inputs/
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import fixture_io  # noqa: E402

f32, i32 = np.float32, np.int32
N = 2048


def _unit(v):
    return (v / np.linalg.norm(v, axis=1, keepdims=True)).astype(f32)


def inputs():
    rng = np.random.default_rng(20261004)
    a = {}
    # ZSobol (sampler/sobol.jl:269-309): pixel coordinates and sample indices are passed VERBATIM (1-based, Q1); film 64 x 64, log2_spp from 4096
    a["sobol_px"] = rng.integers(1, 65, N).astype(i32)
    a["sobol_py"] = rng.integers(1, 65, N).astype(i32)
    a["sobol_sidx"] = rng.integers(1, 4097, N).astype(i32)
    a["sobol_dim"] = rng.integers(0, 70, N).astype(i32)
    # K1 (volpath.jl:143-181): pixels of the 64 x 64 film of test/volpath_integration.jl, sample indices 1 .. 64
    a["cam_px"] = rng.integers(1, 65, N).astype(i32)
    a["cam_py"] = rng.integers(1, 65, N).astype(i32)
    a["cam_sidx"] = rng.integers(1, 65, N).astype(i32)
    # uplift (spectral/uplift.jl:255-308, 514-566): colours incl. grey, saturated and > 1 (the unbounded / illuminant forms), wavelengths 360 .. 830
    rgb = rng.random((N, 3)) ** 2 * 1.5
    rgb[:64] = rng.random((64, 1))                      # grey: c0 = c1 = 0
    rgb[64:96] = np.eye(3)[rng.integers(0, 3, 32)]      # pure primaries
    rgb[96:128] = 0.0
    a["uplift_rgb"] = rgb.astype(f32)
    a["uplift_lambda"] = (360 + 470 * rng.random((N, 4))).astype(f32)
    # BSDFs (materials/spectral-eval.jl): directions in world space, a random shading normal, wavelengths, the 2-D and 1-D draws
    a["bsdf_wo"] = _unit(rng.normal(size=(N, 3)))
    a["bsdf_wi"] = _unit(rng.normal(size=(N, 3)))
    a["bsdf_ns"] = _unit(rng.normal(size=(N, 3)))
    a["bsdf_lambda"] = (360 + 470 * rng.random((N, 4))).astype(f32)
    a["bsdf_u"] = rng.random((N, 2)).astype(f32)
    a["bsdf_uc"] = rng.random(N).astype(f32)
    # light BVH + light sampling (lights/bvh-light-sampler.jl:105-232, physical-wavefront/lights.jl:39-297): shading points inside the box
    a["light_p"] = (rng.random((N, 3)) * np.array([1.8, 1.8, 1.8]) + np.array([-0.9, 0.05, -0.9])).astype(f32)
    a["light_n"] = _unit(rng.normal(size=(N, 3)))
    a["light_u1"] = rng.random(N).astype(f32)
    a["light_u2"] = rng.random((N, 2)).astype(f32)
    a["light_lambda"] = (360 + 470 * rng.random((N, 4))).astype(f32)
    # NanoVDB (volpath/nanovdb.jl:315-388, 602-858): a dense 40 x 24 x 20 field, 45 % empty (whole 8^3 blocks empty too), and index queries
    # in and around its index bounding box
    d = rng.random((40, 24, 20)).astype(f32)
    d[d < 0.45] = 0.0
    d[8:16, :, :] = 0.0
    d[:, 16:, 8:] = 0.0
    a["nvdb_density"] = d                                # [nx, ny, nz] in NumPy order: Julia reads it as an Array{Float32,3} of size (nz, ny, nx) and permutes
    a["nvdb_ijk"] = np.stack([rng.integers(-3, 44, N), rng.integers(-3, 28, N), rng.integers(-3, 24, N)], axis=1).astype(i32)
    # world points in and around the medium's bounds ((-0.5, 0, -0.3) .. (0.5, 0.6, 0.2)): sample_point's trilinear lookup (nanovdb.jl:400-483)
    lo, hi = np.array([-0.5, 0.0, -0.3]), np.array([0.5, 0.6, 0.2])
    a["nvdb_p"] = (lo - 0.1 * (hi - lo) + 1.2 * (hi - lo) * rng.random((N, 3))).astype(f32)
    # a 16 x 16 equal-area environment map [v, u, rgb]: smooth sky, dark ground, one hot texel (the Distribution2D's binary searches see a spike)
    env = (0.2 + 0.8 * rng.random((16, 16, 3))).astype(f32)
    env[8:, :, :] *= f32(0.1)
    env[3, 11, :] = (40.0, 36.0, 30.0)
    a["env_rgb"] = env
    return a


if __name__ == "__main__":
    out = os.path.join(HERE, "reference_inputs")
    fixture_io.write_set(out, inputs())
    print("wrote", out, sum(os.path.getsize(os.path.join(out, f)) for f in os.listdir(out)), "bytes")

"""CPU-side checks (-m "not gpu"): the C-ABI library loads and exports every symbol the header declares,
the ctypes mirror matches the C struct sizes, host-side scene flattening follows the reference's rules,
and the multi-GPU sample sharding + film reduce works over gloo with world_size 2."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol(hk):
    if not os.path.isfile(hk.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    hdr = open(os.path.join(ROOT, "include", "hikari_mi355x.h")).read()
    declared = sorted(set(re.findall(r"\b(hk_[a-z0-9_]+)\s*\(", hdr)))
    assert set(declared) == set(hk._abi.EXPORTED_SYMBOLS)
    lib = C.CDLL(hk.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name


def test_ctypes_mirror_matches_c_layout(hk, tmp_path):
    names = ["hk_texture", "hk_tex_rgba", "hk_tex_f32", "hk_material", "hk_pl_spectrum", "hk_medium_interface", "hk_tri_meta",
             "hk_light", "hk_envmap", "hk_medium", "hk_scene_desc", "hk_tables", "hk_integrator_params", "hk_camera", "hk_stats"]
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include "hikari_mi355x.h"\nint main(){' +
                   "".join('printf("%s %%zu\\n", sizeof(%s));' % (n, n) for n in names) + "return 0;}")
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    out = dict(l.split() for l in subprocess.check_output([str(exe)]).decode().splitlines())
    for n in names:
        assert int(out[n]) == C.sizeof(getattr(hk._abi, n)), n


def test_missing_gpu_or_library_fails_loudly(hk):
    """No CPU fallback: without a GPU hk_ctx_create must return an error, never render on the host."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    L = hk._lib.lib()
    h = C.c_void_p()
    st = L.hk_ctx_create(0, None, C.byref(h))
    if st == 0:                      # a device the HIP runtime sees even though torch does not
        L.hk_ctx_destroy(h)
        pytest.skip("GPU present")
    assert st != 0 and L.hk_last_error()
    with pytest.raises(hk.HikariMI355XError):
        from hikari_jl_amd import scenes
        s, film, cam = scenes.single_triangle(8, 6)
        hk.VolPath(samples=1)(s, film, cam)


def test_scene_flattening_rules(hk):
    from hikari_jl_amd import geometry as G
    s = hk.Scene()
    white = hk.MatteMaterial(Kd=hk.RGBSpectrum(0.7))
    s.push(hk.PointLight((0, 1, 0), hk.RGBSpectrum(5.0)))
    a = s.push(G.rect3f((0, 0, 0), (1, 1, 1)), white)
    b = s.push(G.rect3f((2, 0, 0), (1, 1, 1)), white)                       # same material object -> new record, like push!
    em = hk.MediumInterface(hk.MatteMaterial(), emission=hk.Emissive(Le=hk.RGBSpectrum(1.0), scale=2.0))
    s.push(G.quad((0, 2, 0), (1, 2, 0), (1, 2, 1), (0, 2, 1)), em, transform=G.translate((0, 5, 0)))
    s.push(hk.AmbientLight(hk.RGBSpectrum(0.1)))
    d = s.desc
    assert d.n_triangles == 26 and d.n_lights == 4
    kinds = [d.lights[i].kind for i in range(4)]
    A = hk._abi
    assert kinds == [A.HK_LIGHT_POINT, A.HK_LIGHT_DIFFUSE_AREA, A.HK_LIGHT_DIFFUSE_AREA, A.HK_LIGHT_AMBIENT]   # type-slot order
    # area-light flat index = length(scene.lights) at push time (scene-mesh.jl:127-128)
    assert [d.meta[24].arealight_flat_idx_1based, d.meta[25].arealight_flat_idx_1based] == [2, 3]
    assert d.meta[0].primitive_index == 1 and d.meta[12].primitive_index == 1
    # quirk Q18: light vertices are the UN-transformed mesh, geometry is transformed
    assert abs(d.lights[1].v[1] - 2.0) < 1e-6 and abs(d.positions[24 * 9 + 1] - 7.0) < 1e-6
    assert abs(d.lights[1].area - 0.5) < 1e-6 and d.lights[1].scale == 2.0
    # scale quirk Q4: position-first ctor keeps scale 1; RGB ctor bakes an illuminant with 1/10567
    assert d.lights[0].scale == 1.0
    pl = hk.PointLight.from_rgb((50, 50, 50), (0, 0, 0))
    assert abs(pl.scale - 1 / 10567.0) < 1e-9 and abs(pl.i.scale - 100.0) < 1e-4 and pl.i.poly[0] == 0.0


def test_camera_matches_reference_conventions(hk, oracle):
    """test/film.jl camera test checks ray-direction *ordering* across the raster; same here via the oracle."""
    film = hk.Film((64, 48))
    cam = hk.PerspectiveCamera((0, 0, -5), (0, 0, 0), film, fov=45.0)
    p = hk.integrator_params(filter=hk.BoxFilter((1e-6, 1e-6)))
    px = np.array([1, 64, 1, 64], np.int32)
    py = np.array([1, 1, 48, 48], np.int32)
    r = oracle.camera_samples(p, cam, 64, 48, px, py, np.ones(4, np.int32))
    d = r[:, 12:15]
    assert np.allclose(np.linalg.norm(d, axis=1), 1, atol=1e-6) and (d[:, 2] > 0).all()
    assert d[0, 0] < d[1, 0] and d[2, 0] < d[3, 0]            # x increases with px
    assert d[0, 1] > d[2, 1]                                   # film y = height - y + 1.5: pixel row y=1 is the TOP of the image (Q2)


def test_matrix_camera_equals_perspective_camera(hk, oracle):
    """MatrixCamera (camera/matrix.jl) fed the view / projection a PerspectiveCamera is built from gives the same rays.  The
    PerspectiveCamera mirror keeps the camera looking along -z in its own space (perspective.jl:109), i.e. pbrt's matrices with the
    z axis reversed: the two records differ by exactly those signs and the pinhole rays are bit-identical."""
    from hikari_jl_amd import geometry as G
    film = hk.Film((40, 30))
    pc = hk.PerspectiveCamera((1, 2, -5), (0, 0.5, 0), film, fov=35.0)
    view = G.look_at((1, 2, -5), (0, 0.5, 0), (0, 1, 0))
    proj = G.perspective(35.0, 0.01, 1000.0)
    mc = hk.MatrixCamera(view, proj, film)
    flip = np.diag([1.0, 1.0, -1.0, 1.0]).astype(np.float32)
    assert np.allclose(flip @ mc.raster_to_camera, pc.raster_to_camera, rtol=1e-6, atol=1e-7)
    assert np.allclose(mc.camera_to_world @ flip, pc.camera_to_world, rtol=1e-6, atol=1e-7)
    assert pc._apply_point(pc.raster_to_camera, (20, 15, 0))[2] < 0            # the camera looks along -z in camera space
    assert np.allclose(mc.dx_camera, pc.dx_camera, atol=1e-7) and mc.A > 0
    p = hk.integrator_params()
    rng = np.random.default_rng(4)
    px, py = rng.integers(1, 41, 50).astype(np.int32), rng.integers(1, 31, 50).astype(np.int32)
    idx = rng.integers(1, 64, 50).astype(np.int32)
    a = oracle.camera_samples(p, pc, 40, 30, px, py, idx)
    b = oracle.camera_samples(p, mc, 40, 30, px, py, idx)
    assert np.allclose(a, b, rtol=1e-5, atol=1e-6)
    rec = mc.record()
    assert rec.lens_radius == 0.0 and rec.shutter_open == 0.0 and rec.shutter_close == 1.0
    # resolution may be given as a (w, h) pair, as the reference's Point2f
    mc2 = hk.MatrixCamera(view.astype(np.float64), proj.astype(np.float64), (40, 30))
    assert np.array_equal(mc2.raster_to_camera, mc.raster_to_camera)


WORKER = r'''
import os, sys
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "oracle"))
import numpy as np, torch, torch.distributed as dist
import hikari_jl_amd as hk
from hikari_jl_amd import scenes, distributed as hd
import oracle
dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
rank, world = dist.get_rank(), dist.get_world_size()
s, film, cam = scenes.cornell_box(24, 24, light="point", spheres=False)
p = hk.integrator_params(max_depth=4, samples=6)
first, count, stride = hd.shard_samples(6, rank, world)
osc = oracle.OracleScene(s)
acc, _ = osc.render(p, cam, 24, 24, count, first=first, stride=stride)     # this rank's strided sample set (checker stands in for the GPU)
t = torch.from_numpy(acc)
hd.reduce_film(t, root=0)
if rank == 0:
    full, _ = osc.render(p, cam, 24, 24, 6)
    assert np.allclose(t.numpy(), full, rtol=1e-5, atol=1e-6), "sharded film != single film"
    print("OK")
dist.destroy_process_group()
'''


def test_two_rank_gloo_sample_sharding_and_film_reduce(tmp_path):
    script = tmp_path / "w.py"
    script.write_text(WORKER % {"root": ROOT})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.PIPE) for r in range(2)]
    outs = [p.communicate(timeout=300) for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert b"OK" in outs[0][0]


def test_shard_samples_partition(hk):
    from hikari_jl_amd import distributed as hd
    for total in (1, 7, 256):
        for world in (1, 2, 4, 8):
            seen = []
            for r in range(world):
                first, count, stride = hd.shard_samples(total, r, world)
                seen += [first + k * stride for k in range(count)]
            assert sorted(seen) == list(range(1, total + 1))


def test_postprocess_oracle_closed_forms(hk, oracle):
    """postprocess_kernel! (src/postprocess.jl:185-250) restated in the oracle, against the formulas written out in numpy:
    ACES / Reinhard / extended Reinhard / Uncharted 2 / filmic / linear clamp, gamma, sensor imaging ratio, Bradford white
    balance (identity at the D65 temperature ~6504 K), and the 3x3 escaped-ray mask."""
    from hikari_jl_amd.postprocess import make_params, compute_white_balance_matrix
    rng = np.random.default_rng(2)
    fb = (rng.random((6, 5, 3)) * 3).astype(np.float32)
    x = fb.astype(np.float64)
    out = oracle.postprocess(make_params(exposure=1.5, tonemap="aces", gamma=2.2), fb)
    e = x * 1.5
    ref = np.clip((e * (2.51 * e + 0.03)) / (e * (2.43 * e + 0.59) + 0.14), 0, 1) ** (1 / 2.2)
    assert np.allclose(out, ref, rtol=2e-5, atol=1e-6)
    out = oracle.postprocess(make_params(tonemap="reinhard", gamma=None), fb)
    lum = 0.2126 * x[..., 0] + 0.7152 * x[..., 1] + 0.0722 * x[..., 2]
    assert np.allclose(out, np.clip(x / (1 + lum)[..., None], 0, 1), rtol=2e-5, atol=1e-6)
    out = oracle.postprocess(make_params(tonemap="reinhard_extended", gamma=None, white_point=2.0), fb)
    assert np.allclose(out, np.clip(x * ((1 + lum / 4.0) / (1 + lum))[..., None], 0, 1), rtol=2e-5, atol=1e-6)
    out = oracle.postprocess(make_params(tonemap=None, gamma=None, sensor=hk.FilmSensor(iso=50, exposure_time=0.5)), fb)
    assert np.allclose(out, np.clip(x * 0.25, 0, 1), rtol=1e-6)
    f = lambda v: (np.maximum(v - 0.004, 0) * (6.2 * np.maximum(v - 0.004, 0) + 0.5)) / (np.maximum(v - 0.004, 0) * (6.2 * np.maximum(v - 0.004, 0) + 1.7) + 0.06)
    assert np.allclose(oracle.postprocess(make_params(tonemap="filmic", gamma=None), fb), f(x), rtol=2e-5, atol=1e-6)
    u2 = lambda v: ((v * (0.15 * v + 0.05) + 0.004) / (v * (0.15 * v + 0.5) + 0.06)) - 0.02 / 0.3
    assert np.allclose(oracle.postprocess(make_params(tonemap="uncharted2", gamma=None), fb), np.clip(u2(2 * x) / u2(11.2), 0, 1), rtol=5e-5, atol=1e-6)
    M = compute_white_balance_matrix(6504.0)
    assert np.allclose(M, np.eye(3), atol=4e-2)            # the Planckian locus at 6504 K is close to, not on, the D65 white point
    M5 = compute_white_balance_matrix(5000.0)
    assert M5[2, 2] > 1.05 and M5[0, 0] < 1.0            # warm source -> boost blue, cut red
    depth = np.full((6, 5), 2.0, np.float32)
    depth[:, 3:] = np.inf
    out = oracle.postprocess(make_params(tonemap=None, gamma=None, background=(0.2, 0.4, 0.6)), fb, depth)
    assert np.allclose(out[:, 4], [0.2, 0.4, 0.6], atol=1e-6)                                      # fully escaped neighbourhood
    assert np.allclose(out[:, 0], np.clip(fb[:, 0], 0, 1), atol=1e-6)                              # untouched far from the edge
    assert not np.allclose(out[:, 2], np.clip(fb[:, 2], 0, 1), atol=1e-3)                          # blended at the silhouette


def test_medium_presets_table_and_constructors(hk):
    """volpath/media.jl:1769-1829 (40 named media) and the constructors :1874-2032: values as float32, scale multiplies both
    coefficients, Smoke / Fog follow their density / albedo formulas, unknown names are errors."""
    f32 = np.float32
    from hikari_jl_amd import media_presets as MP
    assert len(MP._presets()) == 40
    p = hk.get_medium_preset("Wholemilk")
    assert p["sigma_s"] == [2.55, 3.21, 3.77] and p["sigma_a"] == [0.0011, 0.0024, 0.014]
    assert hk.get_medium_preset("Spectralon")["sigma_a"] == [0.0, 0.0, 0.0]
    m = hk.Milk(scale=0.5, g=0.8)
    assert m.sigma_s.c[:3] == tuple(float(f32(v) * f32(0.5)) for v in (2.55, 3.21, 3.77)) and m.g == float(f32(0.8))
    assert hk.Coffee().sigma_a.c[:3] == tuple(float(f32(v)) for v in (4.80, 6.58, 8.85))
    assert hk.Juice("grape").sigma_s.c[:3] == (float(f32(5.4e-5)), 0.0, 0.0)
    assert hk.Wine("merlot", scale=2).sigma_a.c[:3] == tuple(float(f32(v) * f32(2)) for v in (0.116, 0.252, 0.294))
    assert hk.SubsurfaceMedium("Ketchup").sigma_a.c[1] == float(f32(0.97))
    s = hk.Smoke(density=2.0, albedo=0.75)
    assert s.sigma_s.c[0] == 1.5 and s.sigma_a.c[0] == 0.5 and s.sigma_s.c[0] == s.sigma_s.c[2]
    fog = hk.Fog(density=0.3)
    assert fog.sigma_s.c[0] == float(f32(0.3)) and fog.sigma_a.c[0] == float(f32(0.3) * f32(0.001))
    for bad in (lambda: hk.get_medium_preset("Mercury"), lambda: hk.Juice("mango"), lambda: hk.Wine("rioja")):
        with pytest.raises(KeyError):
            bad()
    rec = hk._abi.hk_medium()
    m.fill_record(rec, [])
    assert rec.kind == hk._abi.HK_MEDIUM_HOMOGENEOUS and rec.sigma_s[2] == f32(3.77) * f32(0.5)


@pytest.mark.parametrize("n_tris", [1, 4, 5, 257, 1934, 60000])
def test_bvh_builder_invariants(n_tris):
    """bvh_build.cpp on the host alone (g++, no GPU): breadth-first node numbering — the first nodes of the array are the top of the tree,
    which is what k_trace_lean's LDS node cache holds — every node and triangle referenced exactly once, leaves of <= 4 inside their
    child box, depth within the traversal stack (tests/native/bvh_order_check.cpp)."""
    import tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(tempfile.gettempdir(), "hk_bvh_order_check_%d" % os.getuid())
    src = [os.path.join(root, "tests", "native", "bvh_order_check.cpp"), os.path.join(root, "hikari.jl_amd", "csrc", "bvh_build.cpp")]
    if not os.path.exists(exe) or any(os.path.getmtime(f) > os.path.getmtime(exe) for f in src):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-I", os.path.join(root, "include"), "-I", os.path.join(root, "hikari.jl_amd", "csrc")] + src + ["-o", exe])
    r = subprocess.run([exe, str(n_tris), "11"], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.startswith("ok"), (r.stdout, r.stderr)

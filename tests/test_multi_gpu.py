"""Multi-GPU behind the C-ABI: pixel-tile sharding (hk_render_tile) and the in-library RCCL film
reduce (hk_comm_* / hk_film_reduce).  One GPU is all the test box has, so the -m gpu part runs the partitions sequentially on
cuda:0 (every rank's work is independent by construction) and RCCL as a 1-rank communicator; the 2-rank exchange itself is
covered on CPU over gloo with the oracle standing in for the device (the product never runs on the CPU)."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_tiles_partition(hk):
    from hikari_jl_amd import distributed as hd
    for (w, h) in ((800, 800), (1024, 1024), (37, 23), (8, 8), (5, 3)):
        for world in (1, 2, 3, 4, 8):
            covered = np.zeros((h, w), int)
            for r in range(world):
                x0, y0, x1, y1 = hd.shard_tiles(w, h, r, world)
                assert 0 <= x0 <= x1 <= w and 0 <= y0 <= y1 <= h and (y0 % 8 == 0)
                covered[y0:y1, x0:x1] += 1
            assert (covered == 1).all(), (w, h, world)


WORKER = r'''
import os, sys
sys.path[:0] = [%(root)r, os.path.join(%(root)r, "oracle")]
import numpy as np, torch, torch.distributed as dist
import hikari_jl_amd as hk, oracle
from hikari_jl_amd import scenes, distributed as hd
dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
rank, world = dist.get_rank(), dist.get_world_size()
w, h = 24, 24
s, film, cam = scenes.cornell_box(w, h, light="point", spheres=False)
p = hk.integrator_params(max_depth=4, samples=3)
osc = oracle.OracleScene(s)
full, _ = osc.render(p, cam, w, h, 3)
x0, y0, x1, y1 = hd.shard_tiles(w, h, rank, world)
mask = np.zeros((h, w), bool)
mask[y0:y1, x0:x1] = True                      # this rank's tile: every other pixel of its film stays zero
mine = np.concatenate([(full[:3 * w * h].reshape(h, w, 3) * mask[..., None]).reshape(-1), (full[3 * w * h:].reshape(h, w) * mask).reshape(-1)]).astype(np.float32)
t = torch.from_numpy(mine.copy())
hd.reduce_film(t, root=0)
if rank == 0:
    assert np.array_equal(t.numpy(), full), "tile-sharded film != single film"
    print("OK")
dist.destroy_process_group()
'''


def test_two_rank_gloo_tile_sharding_and_film_reduce(tmp_path):
    script = tmp_path / "w.py"
    script.write_text(WORKER % {"root": ROOT})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29547", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.PIPE) for r in range(2)]
    outs = [p.communicate(timeout=300) for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert b"OK" in outs[0][0]


@pytest.mark.gpu
def test_tile_sharding_is_bit_exact(hk):
    """hk_render_tile: 1, 2, 4 and 8 horizontal bands (plus an arbitrary rectangle split) rendered into separate zero-initialised
    films sum to exactly the accumulators of the whole-film render — same samples, same order, per pixel — on a ragged film
    (not a multiple of the 8x8 tile) with a medium in the scene (ticketed segments) and on Cornell (static segments)."""
    from hikari_jl_amd import distributed as hd
    from hikari_jl_amd import scenes
    # 16 samples per call: the call reads the sampler's sample-bit table (built per tile origin), 4: it hashes the digits
    for which, (w, h), ns in (("cornell", (67, 45), 4), ("cornell", (67, 45), 16), ("integration", (40, 36), 4)):
        s, _, _ = (scenes.cornell_box(w, h, light="area") if which == "cornell" else scenes.integration_test_scene(w, h))
        film = hk.Film((w, h))
        cam = hk.PerspectiveCamera((0, 1, -3.5), (0, 1, 0), film, fov=40.0)
        kw = dict(max_depth=5, samples=ns)
        vp = hk.VolPath(**kw)
        vp(s, film, cam)
        whole = vp.read_accumulators(film)
        rays = int(vp.stats().rays_closest)
        vp.close()
        splits = [[hd.shard_tiles(w, h, r, world) for r in range(world)] for world in (1, 2, 4, 8)]
        splits.append([(0, 0, 19, h), (19, 0, w, 17), (19, 17, w, h)])
        for tiles in splits:
            total = np.zeros_like(whole)
            nrays = 0
            for tile in tiles:
                f = hk.Film((w, h))
                v = hk.VolPath(**kw)
                v._ensure(f)
                v.clear()
                v.reset_stats()
                v.render_samples(s, f, cam, ns, tile=tile, readback=False)
                part = v.read_accumulators(f)
                x0, y0, x1, y1 = tile
                m = np.zeros((h, w), bool)
                m[y0:y1, x0:x1] = True
                assert not part[3 * w * h:].reshape(h, w)[~m].any()              # nothing outside the tile is touched
                total += part
                nrays += int(v.stats().rays_closest)
                v.close()
            if which == "cornell":
                assert np.array_equal(total, whole), (which, tiles)
                assert nrays == rays
            else:   # delta tracking re-seeds from ray bits: identical here too, the same path runs whatever tile it is rendered in
                assert np.array_equal(total, whole), (which, tiles)


@pytest.mark.gpu
def test_rccl_film_reduce_in_library(hk):
    """hk_comm_create over the local device and hk_comm_create_rank (world 1) + hk_film_reduce: RCCL is loaded, the communicator
    comes up, the reduce runs on the context's stream in place and leaves the single rank's accumulators untouched; precision
    and size mismatches are rejected with a message."""
    from hikari_jl_amd import scenes
    w = h = 32
    s, film, cam = scenes.cornell_box(w, h, light="area")
    ctx = hk.Context.get(0)
    vp = hk.VolPath(max_depth=4, samples=4)
    vp(s, film, cam)
    before = vp.read_accumulators(film)
    comm = hk.Comm.local([ctx])
    comm.reduce_films([vp], root=0)
    vp.sync()
    assert np.array_equal(vp.read_accumulators(film), before)
    comm.close()
    uid = hk.Comm.unique_id()
    assert len(uid) == 128 and any(uid)
    comm = hk.Comm.rank(ctx, uid, 0, 1)
    vp.render_samples(s, film, cam, 4, readback=False)        # reduce is ordered after the render on the same stream
    comm.reduce_films([vp], root=0)
    vp.sync()
    after = vp.read_accumulators(film)
    assert np.allclose(after[3 * w * h:], 2 * before[3 * w * h:], rtol=1e-6)
    L = hk._lib.lib()
    films = (C.c_void_p * 1)(vp._film[0])
    assert L.hk_film_reduce(comm.h, films, 1, 3) == hk._abi.HK_ERR_INVALID and b"root" in L.hk_last_error()
    assert L.hk_film_reduce(comm.h, films, 2, 0) == hk._abi.HK_ERR_INVALID
    comm.close()
    vp.close()


def test_bench_launcher_propagates_a_failing_rank():
    """`bench.py --gpus 2` without a launcher starts the two ranks itself.  Here (no GPU) both ranks fail at torch.cuda.set_device;
    the parent must come back non-zero, name the rank, and leave no child behind."""
    env = dict(os.environ, HK_BENCH_LAUNCH_TIMEOUT="240")
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    import torch
    if torch.cuda.is_available():
        pytest.skip("needs a box without a GPU: the ranks are meant to fail")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"],
                       env=env, capture_output=True, timeout=600)
    assert r.returncode != 0
    assert b"failed" in r.stderr


@pytest.mark.gpu
def test_bench_two_ranks_through_the_launcher(hk):
    """The form the driver runs: `python bench.py --gpus 2` (no torchrun).  Both ranks share cuda:0 over gloo (HK_BENCH_SINGLE_DEVICE);
    the parent forwards rank 0's line: n_gpus == 2, strong scaling (128 of the frame's 256 spp... here 8 of 16 per rank), the
    reduced film equals torch.distributed's reduce, and the ray count is the one-GPU frame's."""
    import json
    env = dict(os.environ, HK_BENCH_SINGLE_DEVICE="1", HK_BENCH_LAUNCH_TIMEOUT="900")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    base = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--spp", "16"]
    r2 = subprocess.run(base + ["--gpus", "2"], env=env, capture_output=True, timeout=1200)
    assert r2.returncode == 0, r2.stderr[-3000:]
    d2 = json.loads(r2.stdout.decode().strip().splitlines()[-1])
    assert d2["n_gpus"] == 2 and d2["scaling"] == "strong"
    assert d2["config"]["spp_per_frame"] == 16 and d2["config"]["spp_per_rank"] == 8
    assert d2["config"]["reduce_matches_torch_distributed"] is True
    r1 = subprocess.run(base + ["--gpus", "1"], env=dict(env, HK_BENCH_SINGLE_DEVICE="0"), capture_output=True, timeout=1200)
    assert r1.returncode == 0, r1.stderr[-3000:]
    d1 = json.loads(r1.stdout.decode().strip().splitlines()[-1])
    assert d1["n_gpus"] == 1
    # the two ranks together cast the rays of the one-GPU frame (same sample set; the hashed decisions do not depend on the sharding)
    assert abs(d2["rays"]["total_all_ranks"] - d1["rays"]["total_all_ranks"]) <= 1e-3 * d1["rays"]["total_all_ranks"]
    w = subprocess.run(base + ["--gpus", "2", "--scaling", "weak"], env=env, capture_output=True, timeout=1200)
    assert w.returncode == 0, w.stderr[-3000:]
    dw = json.loads(w.stdout.decode().strip().splitlines()[-1])
    assert dw["scaling"] == "weak" and dw["config"]["spp_per_frame"] == 32 and dw["config"]["spp_per_rank"] == 16

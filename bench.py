#!/usr/bin/env python3
"""bench.py — Mrays/s of the VolPath hot path on BASELINE.json configs[1]:
Cornell box (diffuse + area light), 800x800, depth 8, 256 spp, 1 x MI355X.

A "step" is one wavefront pass of the hot path over one batch of synthetic input: SPP_PER_STEP (=256)
samples of every pixel of the 800x800 frame carried through the 8-bounce loop (164 M paths in flight, ~72 GB
of path state of the 288 GB of HBM: the deeper bounces of a pass only keep the chip busy when the pass starts
with hundreds of paths per resident lane — 5.9 G rays/s at 32 spp per pass, 6.3 at 64, 6.45 at 128, 6.56 at
256).  The default single step is exactly the 256-spp frame, so `seconds_to_256spp` is the timed region itself
(--spp-per-step changes the batch; --steps defaults to 256 / spp-per-step).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

N > 1: one process per GPU; rank g renders sample indices g+1, g+1+N, ... (weak scaling: K steps each), then
ONE RCCL sum-reduce of the film accumulators to rank 0 inside the timed region — the library's own hk_film_reduce
(ncclReduce on the render stream; torch.distributed only carries the 128-byte communicator id and the final
statistics, and its reduce serves as the untimed cross-check of the result).  value = rays of all ranks /
max-over-ranks time.  The scene, BVH and film live in HBM before the timed region starts.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

W, H, DEPTH, SPP_PER_STEP, FULL_SPP = 800, 800, 8, 256, 256
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
S_NODE, B_TRI, B_HIT, B_RAY_IN, B_HIT_OUT = 64, 36, 96, 32, 16   # SURVEY.md §8(d) algorithmic bytes per cast
S_STATE = 104                  # compact path state, read + written once per path vertex


def main():
    global SPP_PER_STEP
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="default: the full 256-spp frame (256 / spp-per-step)")
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-spp", type=int, default=0, help="samples per pixel of the CPU baseline sample (0 = about 15 s of work)")
    ap.add_argument("--spp-per-step", type=int, default=SPP_PER_STEP, help="samples of every pixel in flight per wavefront pass (one step)")
    ap.add_argument("--config", default="cornell", choices=["cornell", "cloud", "sky", "manylight"],
                    help="cornell = BASELINE configs[1] (the bench line); sky = configs[2] stand-in (glass sphere + gold slab + env map + sun, depth 12); "
                         "cloud = configs[3] stand-in (synthetic NanoVDB cloud, 1024x1024, depth 32); manylight = configs[4] stand-in (10^6 triangles, "
                         "5*10^4 area lights, 1024x1024, depth 8)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # HK_BENCH_SINGLE_DEVICE=1 (test hook): every rank shares cuda:0 and the process group is gloo, so the N > 1 code path can be
    # exercised on a 1-GPU box.  The driver's runs never set it: one process per GPU over RCCL ("nccl").
    single_device = os.environ.get("HK_BENCH_SINGLE_DEVICE") == "1"
    if args.gpus > 1 or world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group("gloo" if single_device else "nccl", rank=rank, world_size=world)
    if single_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)

    import hikari_jl_amd as hk
    from hikari_jl_amd import distributed as hd
    from hikari_jl_amd import scenes

    global W, H, DEPTH
    SPP_PER_STEP = args.spp_per_step
    if args.steps is None:
        args.steps = max(FULL_SPP // SPP_PER_STEP, 1)
    if args.config == "cloud":
        W, H, DEPTH = 1024, 1024, 32
        mres = int(os.environ.get("HK_CLOUD_MAJORANT", "32"))
        scene, film, cam = scenes.cloud_scene(W, H, "nanovdb", res=(256, 256, 128), sigma_scale=620.0 / 4, majorant_res=(mres, mres, mres))
        workload = "synthetic BOMEX-like NanoVDB cloud (256x256x128, delta tracking), 1024x1024, VolPath depth 32, %d spp per step" % SPP_PER_STEP
    elif args.config == "sky":
        W, H, DEPTH = 800, 800, 12
        scene, film, cam = scenes.sky_scene(W, H, env_res=512)
        workload = "README scene: glass sphere + Gold(roughness=0.01) slab + Hosek-Wilkie sun-sky (512^2 equal-area env map + SunLight), 800x800, VolPath depth 12, %d spp per step" % SPP_PER_STEP
    elif args.config == "manylight":
        W, H, DEPTH = 1024, 1024, 8
        scene, film, cam = scenes.many_light_scene(W, H)
        workload = "synthetic many-light barrel (10^6 triangles, ~5*10^4 area lights in the light BVH), 1024x1024, VolPath depth 8, %d spp per step" % SPP_PER_STEP
    else:
        scene, film, cam = scenes.cornell_box(W, H, light="area")
        workload = "Cornell box (diffuse + area light), 800x800, VolPath depth 8, %d spp per step" % SPP_PER_STEP
    n_pix = W * H
    # film accumulators live in a torch tensor so torch.distributed (RCCL) can reduce them in place
    accum = torch.zeros(4 * n_pix, dtype=torch.float32, device="cuda")
    # `samples` only sizes the ZSobol index (log2 of max(samples, 4096)): cover every sample index this run touches
    vp = hk.VolPath(max_depth=DEPTH, samples=max(FULL_SPP, SPP_PER_STEP * world), samples_per_pass=SPP_PER_STEP,
                    device=local_rank)
    vp.use_external_accumulators(accum.data_ptr())
    vp._ensure(film)
    L = hk._lib.lib()

    def run_steps(first_step, n_steps, readback=False):
        # A step is one frame, the way the reference's `integrator(scene, film, camera)` renders one: clear the film, then
        # SPP_PER_STEP samples per pixel — on this rank the sample indices (rank+1) + world*j, j < SPP_PER_STEP.  Every step
        # renders the same sample indices (a frame loop), so the sampler tables built by the first frame serve the later ones.
        for _ in range(n_steps):
            vp.clear()
            vp.render_samples(scene, film, cam, SPP_PER_STEP, stride=world, first=rank + 1, readback=readback)

    def barrier():
        if world > 1:
            dist.barrier()
        vp.sync()
        torch.cuda.synchronize()

    # ---- the film reduce of N > 1: hk_comm over RCCL inside the library (the C-ABI's own exchange step) ----
    comm = None
    comm_note = None
    if world > 1 and not single_device:
        # The builder's box has one GPU: this path has only ever run as a 1-rank communicator.  If the in-library communicator cannot
        # be set up on every rank (or its first reduce fails, below), ALL ranks fall back to torch.distributed's reduce — the same
        # RCCL ring on the same buffer — and the line says so, rather than the scaling run dying.
        ok = 1
        try:
            uid = [hk.Comm.unique_id() if rank == 0 else None]
        except Exception as e:           # noqa: BLE001
            uid, ok, comm_note = [None], 0, "hk_comm_unique_id: %s" % e
        dist.broadcast_object_list(uid, src=0)        # the launcher's side channel for the 128-byte id
        if uid[0] is None:
            ok = 0
        if ok:
            try:
                comm = hk.Comm.rank(vp._ctx, uid[0], rank, world)
            except Exception as e:       # noqa: BLE001
                ok, comm_note = 0, "hk_comm_create_rank: %s" % e
        flag = torch.tensor([ok], dtype=torch.int32, device="cuda")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 0:
            comm = None
            comm_note = comm_note or "another rank could not create its communicator"

    def reduce_films():
        if comm is not None:
            comm.reduce_films([vp], root=0)
        else:                                         # HK_BENCH_SINGLE_DEVICE test hook: every rank shares cuda:0, gloo stages through the host
            hd.reduce_film(accum, root=0)

    # ---- warmup (untimed): also uploads the scene / builds the BVH ----
    t0 = time.time()
    run_steps(0, max(args.warmup, 1) if args.warmup > 0 else 0)
    barrier()
    setup_s = time.time() - t0
    reduce_check = None
    if world > 1:
        # untimed: RCCL sets up the reduce's channels on first use; and the cross-check of the in-library reduce against
        # torch.distributed's on the warm-up film (same inputs, same ring: identical sums)
        keep = accum.clone()
        if comm is not None:
            ok = 1
            try:
                reduce_films()
                vp.sync()
            except Exception as e:       # noqa: BLE001
                ok, comm_note = 0, "hk_film_reduce: %s" % e
            flag = torch.tensor([ok], dtype=torch.int32, device="cuda")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) == 0:
                comm = None
                comm_note = comm_note or "hk_film_reduce failed on another rank"
                accum.copy_(keep)
                reduce_films()
        else:
            reduce_films()
        barrier()
        mine = accum.clone()
        accum.copy_(keep)
        hd.reduce_film(accum, root=0)
        barrier()
        if rank == 0:
            reduce_check = bool(torch.allclose(mine, accum, rtol=1e-6, atol=1e-7))
    accum.zero_()
    vp.enable_counters(count_nodes=False, time_kernels=False)
    vp.reset_stats()
    barrier()

    # ---- timed region: EXACTLY K steps + (N>1) the film reduce ----
    t0 = time.perf_counter()
    run_steps(0, args.steps)
    if world > 1:
        reduce_films()
    barrier()
    elapsed = time.perf_counter() - t0
    st = vp.stats()
    rays_local = int(st.rays_closest) + int(st.rays_shadow)
    stat_dev = "cpu" if single_device else "cuda"
    tmax = torch.tensor([elapsed], dtype=torch.float64, device=stat_dev)
    rays = torch.tensor([float(rays_local)], dtype=torch.float64, device=stat_dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(rays, op=dist.ReduceOp.SUM)
    elapsed_max = float(tmax.item())
    total_rays = float(rays.item())

    result = None
    if rank == 0:
        # ---- per-kernel-class times: an untimed replay of the same steps with HIP events around every launch, on the launch stream.
        #      With the events on, the library keeps everything on one stream (the timed region above runs the shadow rays of bounce d
        #      beside the traversal of bounce d + 1 on a second stream), so the class times add up to slightly MORE than the timed frame ----
        accum_timed = accum.clone()
        vp.enable_counters(count_nodes=False, time_kernels=True)
        vp.reset_stats()
        run_steps(0, args.steps)
        vp.sync()
        st = vp.stats()
        accum.copy_(accum_timed)
        # ---- roofline of the dominant kernel: counts from an instrumented (untimed) replay of the same steps ----
        timed = dict(trace=st.seconds_trace, shadow=st.seconds_shadow, shade=st.seconds_shade, media=st.seconds_media, other=st.seconds_other)
        launches = dict(trace=int(st.trace_launches), shadow=int(st.shadow_launches), shade=int(st.shade_launches))
        casts = dict(trace=int(st.rays_closest), shadow=int(st.rays_shadow))
        vertices = int(st.path_vertices)
        accum_keep = accum.clone()
        vp.enable_counters(count_nodes=True, time_kernels=False)
        vp.reset_stats()
        run_steps(0, args.steps)
        vp.sync()
        sc = vp.stats()
        accum.copy_(accum_keep)
        vp.enable_counters(False, False)
        # ---- per-kernel-class ceilings: ALGORITHMIC bytes from the counted replay (hk_stats.bytes_algorithmic_*) over the
        #      HIP-event time of that class's launches in the timed region; PMC traffic / L2 hit rate / VALU issue / lane utilisation
        #      from the rocprofv3 passes committed under profiles/ for the same workload (tools/profile_round.sh) ----
        alg = dict(trace=int(sc.bytes_algorithmic_trace), shadow=int(sc.bytes_algorithmic_shadow), shade=int(sc.bytes_algorithmic_shade))
        kname = {"trace": "k_trace", "shadow": "k_shadow", "shade": "k_shade"}

        def committed(stem):
            path = os.path.join(ROOT, "profiles", "%s_%s.json" % (stem, args.config))
            try:
                return json.load(open(path)) if SPP_PER_STEP == 256 else {}     # the PMC passes were taken at the default 256 spp per step
            except (OSError, ValueError):
                return {}

        pmc, util = committed("pmc_traffic"), committed("utilisation")
        rooflines = []
        for cls in ("trace", "shadow", "shade"):
            n_launch = max(launches[cls], 1)
            avg_s = timed[cls] / n_launch
            if timed[cls] <= 0:
                continue
            achieved = alg[cls] / n_launch / avg_s / 1e9
            e = {"kernel": kname[cls], "bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
                 "algorithmic_bytes_per_launch": int(alg[cls] / n_launch), "avg_launch_ms": round(avg_s * 1e3, 4), "launches": n_launch,
                 "seconds": round(timed[cls], 4)}
            t = pmc.get(kname[cls], {})
            e["traffic"] = t.get("hbm_bytes_per_launch")
            if e["traffic"]:
                # the PMC launches average over the same depths as the bench's (whole passes): traffic per launch is comparable
                e["traffic_over_algorithmic"] = round(e["traffic"] / max(alg[cls] / n_launch, 1), 3)
                e["hbm_frac_by_traffic"] = round(e["traffic"] / avg_s / 1e9 / HBM_PEAK_GBS, 4)
            if "l2_hit_rate" in t:
                e["l2_hit_rate"] = t["l2_hit_rate"]
            u = util.get(kname[cls], {})
            for k in ("valu_issue_frac", "lane_util", "wait_frac"):
                if k in u:
                    e[k] = u[k]
            hb, vi = e.get("hbm_frac_by_traffic"), e.get("valu_issue_frac")
            # valu_issue_frac prices a wave64 VALU instruction at the data sheet's 2 cycles; measured on this chip (tools/valu_rate.hip,
            # profiles/r02_valu_rate.txt) v_fma / v_mul / v_mov cost 2.3 - 2.9 and nearly everything else 4.2 - 4.4, so 0.25 - 0.45 in
            # data-sheet units is a saturated issue port for these instruction mixes
            if hb is None or vi is None:
                e["binding"] = "unprofiled on this workload"
            elif hb >= 0.5:
                e["binding"] = "HBM traffic (%.0f %% of the 8 TB/s peak, %.0f %% of the device-copy rate); VALU issue %.0f %% at 2 cycles per instruction" % (100 * hb, 100 * hb * HBM_PEAK_GBS / 5200.0, 100 * vi)
            else:
                e["binding"] = "instruction issue and latency: VALU issue %.0f %% of all cycles at 2 cycles per instruction (this mix costs ~4: profiles/r02_valu_rate.txt) with %.0f %% of the lanes active; HBM traffic %.0f %% of peak" % (100 * vi, 100 * e.get("lane_util", 0), 100 * hb)
            rooflines.append(e)
        # BVH nodes come from L1 / L2, not HBM: the SURVEY 8(d) byte formula is an upper bound of traversal's HBM need, not a ceiling
        for e in rooflines:
            if e["kernel"] in ("k_trace", "k_shadow"):
                e["note"] = "algorithmic bytes count every BVH node / triangle visit at full size; measured HBM traffic is the `traffic` field"
        dom = max(("trace", "shadow", "shade"), key=lambda k: timed[k])
        # measured HBM ceiling of this box beside the nominal peak: device-to-device copy, read + write bytes
        a = torch.empty(1 << 28, dtype=torch.float32, device="cuda")    # 1 GiB
        b = torch.empty_like(a)
        b.copy_(a)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            b.copy_(a)
        e1.record()
        torch.cuda.synchronize()
        copy_gbs = 10 * 2 * a.numel() * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e9
        del a, b
        roofline = dict(next(e for e in rooflines if e["kernel"] == kname[dom]))
        roofline.update({"measured_copy_gbs": round(copy_gbs, 1), "frac_of_measured_copy": round(roofline["achieved"] / copy_gbs, 5),
                         "nodes_per_cast": round(int(sc.trace_nodes) / max(int(sc.rays_closest), 1), 2),
                         "tris_per_cast": round(int(sc.trace_tris) / max(int(sc.rays_closest), 1), 2),
                         "kernel_seconds": {k: round(v, 4) for k, v in timed.items()}})

        # ---- CPU baseline: the oracle (a port, NOT Julia / KernelAbstractions.CPU()) on a bounded sample ----
        cpu = None
        if not args.no_cpu_baseline:
            import oracle
            oracle.build()
            oracle.set_threads(oracle.available_cores())     # affinity mask and cgroup quota, not the host's core count
            osc = oracle.OracleScene(scene)
            p = hk.integrator_params(max_depth=DEPTH, samples=FULL_SPP)
            c0 = time.perf_counter()
            osc.render(p, cam, W, H, 1, first=FULL_SPP)       # untimed: thread team start-up, first-touch of the work arrays
            if args.cpu_spp <= 0:                             # auto: about 15 s of CPU work
                args.cpu_spp = max(1, min(32, int(15.0 / max(time.perf_counter() - c0, 1e-3))))
            c0 = time.perf_counter()
            _, ost = osc.render(p, cam, W, H, args.cpu_spp)
            cdt = time.perf_counter() - c0
            crays = int(ost.rays_closest) + int(ost.rays_shadow)
            cpu = {"value": round(crays / cdt / 1e6, 4), "unit": "Mrays/s", "cores": oracle.max_threads(), "kind": "port",
                   "sample": "%d spp of the same %dx%d depth-%d frame (%.1f s); CPU restatement of Hikari VolPath, not Julia" % (args.cpu_spp, W, H, DEPTH, cdt),
                   "seconds_to_256spp_extrapolated": round(cdt * FULL_SPP / args.cpu_spp, 1)}
            osc.close()

        value = total_rays / elapsed_max / 1e6
        spp_done = SPP_PER_STEP * args.steps * world
        result = {
            "metric": "Mrays/s", "value": round(value, 2), "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed_max / max(args.steps, 1) * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": workload,
                       "resolution": [W, H], "max_depth": DEPTH, "spp_per_step": SPP_PER_STEP, "spp_rendered": spp_done,
                       "triangles": int(scene.desc.n_triangles), "lights": int(scene.desc.n_lights), "parallelism": ("sample-index sharding x%d + in-library RCCL film reduce (hk_film_reduce)" % world) if (comm is not None or world == 1) else ("sample-index sharding x%d + %s" % (world, "gloo film reduce on one device (HK_BENCH_SINGLE_DEVICE test hook)" if single_device else "torch.distributed film reduce (in-library communicator unavailable: %s)" % comm_note)),
                       "reduce_matches_torch_distributed": reduce_check},
            "seconds_timed": round(elapsed_max, 4),
            "seconds_to_256spp": round(elapsed_max * FULL_SPP / spp_done * world, 4) if world == 1 else round(elapsed_max * (FULL_SPP / spp_done), 4),
            "rays": {"closest": int(st.rays_closest), "shadow": int(st.rays_shadow), "total_all_ranks": int(total_rays), "medium_collisions": int(st.medium_collisions)},
            "setup_seconds": round(setup_s, 3),
            "roofline": roofline, "rooflines": rooflines, "cpu_baseline": cpu,
        }
        print(json.dumps(result))
    if comm is not None:
        comm.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    vp.close()
    return result


if __name__ == "__main__":
    main()

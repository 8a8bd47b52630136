#!/usr/bin/env python3
"""bench.py — Mrays/s and seconds per converged frame of the VolPath hot path on BASELINE.json configs[1]:
Cornell box (diffuse + area light), 800x800, depth 8, 256 spp, N x MI355X.

A "step" is ONE FRAME the way the reference's `integrator(scene, film, camera)` renders one: clear the film, FULL_SPP (= 256)
samples of every pixel carried through the 8-bounce loop, and — on N > 1 GPUs — the ONE sum-reduce of the film accumulators onto
rank 0 that finishes the frame.  On one GPU the frame is one wavefront pass (164 M paths in flight, ~72 GB of path state of the
288 GB of HBM: the deeper bounces of a pass only keep the chip busy when the pass starts with hundreds of paths per resident lane —
5.9 G rays/s at 32 spp per pass, 6.3 at 64, 6.45 at 128, 6.56 at 256).

    python bench.py --gpus N --steps K --warmup W            (N > 1 without a launcher: starts the N ranks itself)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

N > 1: one process per GPU; rank g renders the sample indices g+1, g+1+N, ... of ALL pixels (ZSobol is a pure function of
(pixel, sample index, dimension), so the union over ranks is exactly the one-GPU sample set).
  --scaling strong (default: BASELINE.json's metric is the wall-clock to ONE 256-spp frame at 1/2/4/8 GPUs): the frame's 256
      samples are split over the ranks, 256 / N each; `seconds_per_frame` is the measured time of a frame INCLUDING its reduce.
  --scaling weak: every rank renders 256 spp (N x 256 in the reduced film).
The reduce is the library's own hk_film_reduce (ncclReduce over xGMI on the render stream; torch.distributed only carries the
128-byte communicator id, the barriers and the final statistics — over a gloo group on CPU tensors, so that no second stream exists in
the process while frames are timed — and its own RCCL reduce is the cross-check of the result, run after the timed region).
On ONE GPU the default run also reports every other BASELINE.json config (`configs`: the rounds 1-5 Cornell variant, the cloud, the sky, the
many-light scene — seconds per frame, Mrays/s, class times, rooflines, the one-sample-per-call path), each measured by `bench.py --config X`
in a child process of its own BEFORE this process touches the GPU (a second busy hardware queue on the device makes every kernel
launch 50 - 120 us longer: DESIGN.md §5 "two speeds"), and, last, the one-sample-per-call path of the bench scene (`progressive`: hk_render(first = i, n = 1), what an interactive viewer drives).
value = rays of all ranks / max-over-ranks time of the K steps.  Scene, BVH and sampler tables live in HBM before the timed region
starts; `cold_frame_seconds` is a frame that has to rebuild the sample-bit table first (a one-shot render of a new sample range).
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

PROGRESSIVE_CALLS = 64         # --progressive
READBACK_PASS = True
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
COMM_TIMEOUT_S = 180.0         # the collective communicator bring-up may not hang the run


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="frames in the timed region (default 1; Cornell: 20)")
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--scaling", default="strong", choices=["strong", "weak"],
                    help="strong: the frame's samples are split over the ranks (the metric: wall-clock to one 256-spp frame); weak: every rank renders the full sample count")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-spp", type=int, default=0, help="samples per pixel of the CPU baseline sample (0 = about 15 s of work)")
    ap.add_argument("--spp", type=int, default=None, help="samples per pixel of one frame (default: the config's — 256, many-light 512)")
    ap.add_argument("--spp-per-pass", type=int, default=0, help="samples of every pixel in flight per wavefront pass (0 = the library's choice: up to 256)")
    ap.add_argument("--config", default=None, choices=["cornell", "cornell_sphere_box", "cornell_two_spheres", "cloud", "sky", "manylight"],
                    help="default: cornell as the bench line, then ONE warm frame each of cornell_sphere_box, cloud, sky and manylight appended as `configs` (one GPU only; "
                         "--no-extra-configs leaves them out).  cornell = BASELINE configs[1] as SURVEY 8(d) specifies it: two tessellated spheres, 3 782 triangles (cornell_two_spheres: the same); "
                         "cornell_sphere_box = the bench scene of rounds 1-5 (one sphere + one box); sky = configs[2] (glass sphere + gold slab + Hosek-Wilkie sun-sky, depth 12); "
                         "cloud = configs[3] (BOMEX stand-in: worley-fbm NanoVDB cloud field, 1024x1024, depth 32); manylight = configs[4] stand-in "
                         "(10^6 triangles, 5*10^4 area lights, 1024x1024, depth 8, 512 spp)")
    ap.add_argument("--progressive", type=int, default=64, help="calls of the ONE-SAMPLE-PER-CALL path (render!: hk_render(first = i, n = 1), what an interactive "
                                                                 "viewer drives) measured after the timed region and reported as `progressive` (0 = skip)")
    ap.add_argument("--no-readback-pass", action="store_true", help="--progressive without the second pass that reads the frame back after every call (kernel traces)")
    ap.add_argument("--no-extra-configs", action="store_true", help="the default run without the one-frame lines of cloud / sky / manylight")
    ap.add_argument("--detail-file", default=None, help="where the full record goes (default: bench_detail.json in the working directory); stdout carries the compact line only")
    args = ap.parse_args(argv)
    args.extra_configs = args.config is None and not args.no_extra_configs and args.gpus == 1 and not args.spp and not args.spp_per_pass
    args.config = args.config or "cornell"
    global PROGRESSIVE_CALLS, READBACK_PASS
    PROGRESSIVE_CALLS = max(args.progressive, 0)
    READBACK_PASS = not args.no_readback_pass
    return args


def launch_ranks(args):
    """`bench.py --gpus N` without a launcher: start the N ranks as fresh child processes (this parent never touches HIP or
    torch.cuda, and no child is ever re-exec'ed), wait for all of them and forward rank 0's JSON line.  A failing rank fails the run."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out0 = b""
    failed = None
    deadline = time.time() + float(os.environ.get("HK_BENCH_LAUNCH_TIMEOUT", "3000"))
    pending = set(range(args.gpus))
    while pending and failed is None:
        for r in sorted(pending):
            p = procs[r]
            if r == 0:
                try:
                    o, _ = p.communicate(timeout=0.2)
                    out0 += o or b""
                except subprocess.TimeoutExpired:
                    continue
            elif p.poll() is None:
                continue
            pending.discard(r)
            if p.returncode != 0:
                failed = (r, p.returncode)
                break
        if time.time() > deadline:
            failed = (-1, "timeout")
        time.sleep(0.05)
    if failed is not None:
        for p in procs:            # exactly the processes started above
            if p.poll() is None:
                p.kill()
        sys.stderr.write("bench.py: rank %s failed (%s)\n" % failed)
        sys.stdout.write(out0.decode(errors="replace"))
        sys.exit(1)
    sys.stdout.write(out0.decode(errors="replace"))
    sys.stdout.flush()
    sys.exit(0)


def call_with_timeout(fn, seconds, what):
    """Run a (collective) bring-up call in a thread: a rank that never returns must end the run, not hang it."""
    import threading
    box = {}

    def run():
        try:
            box["value"] = fn()
        except BaseException as e:          # noqa: BLE001
            box["error"] = e
    t = threading.Thread(target=run, daemon=True)
    t.start()
    t.join(seconds)
    if t.is_alive():
        sys.stderr.write("bench.py: %s did not return within %.0f s\n" % (what, seconds))
        sys.stderr.flush()
        os._exit(3)
    if "error" in box:
        raise box["error"]
    return box.get("value")


def build_workload(config, scenes):
    """-> scene, film, camera, W, H, depth, spp per frame, workload text (BASELINE.json configs / SURVEY 8d)."""
    if config == "cloud":
        W, H, depth, spp = 1024, 1024, 32, 256
        mres = int(os.environ.get("HK_CLOUD_MAJORANT", "64"))
        scene, film, cam = scenes.bomex_scene(W, H, res=(256, 256, 128), fill=0.05, max_extinction=620.0, majorant_res=(mres, mres, mres))
        workload = ("BOMEX stand-in (the LES data is not in the reference tree): generate_cloud_density's worley-fbm recipe (src/random.jl:149-206, "
                    "pure noise) on 256x256x128, thresholded to 5 %% fill, max extinction 620, NanoVDB + %d^3 majorant grid, sigma_a 0 / sigma_s 1 / g 0.877, "
                    "scene of examples/bomex_cloud_example.jl:53-184 (index-matched glass cube 1.2, floor + two walls, Ambient + Directional), 1024x1024, "
                    "VolPath depth 32" % mres)
    elif config == "sky":
        W, H, depth, spp = 800, 800, 12, 256
        scene, film, cam = scenes.sky_scene(W, H, env_res=512)
        workload = "README scene: glass sphere + Gold(roughness=0.01) slab + Hosek-Wilkie sun-sky (512^2 equal-area env map + SunLight), 800x800, VolPath depth 12"
    elif config == "manylight":
        W, H, depth, spp = 1024, 1024, 8, 512
        scene, film, cam = scenes.many_light_scene(W, H)
        workload = ("synthetic many-light barrel standing in for the absent CMS detector asset (10^6 triangles, ~5*10^4 area lights in the light BVH), "
                    "1024x1024, VolPath depth 8")
    else:
        W, H, depth, spp = 800, 800, 8, 256
        # "two matte boxes/spheres tessellated at 32" (test/volpath_integration.jl:58-62) — the bench line since round 6.
        # Rounds 1-5 led with one sphere + one box (1 934 triangles): still measured, as the `cornell_sphere_box` entry of `configs`.
        objects = os.environ.get("HK_BENCH_CORNELL_OBJECTS", "sphere_box" if config == "cornell_sphere_box" else "two_spheres")
        scene, film, cam = scenes.cornell_box(W, H, light="area", objects=objects)
        workload = ("Cornell box (diffuse + area light; %s, %d triangles), 800x800, VolPath depth 8"
                    % ("two matte spheres tessellated at 32: SURVEY 8(d)" if objects == "two_spheres" else "one tessellated sphere + one box: the rounds 1-5 scene",
                       int(scene.desc.n_triangles)))
    return scene, film, cam, W, H, depth, spp, workload


VALU_CYCLES = 2.0              # data sheet: a wave64 VALU instruction holds a SIMD's issue port for at least 2 cycles -> `valu_issue` (instructions x 2 cycles over the
                               # SIMD-cycles of the launch), a LOWER bound of the pipe's occupancy.  What the hardware itself counts is `valu_busy` = rocprof's VALUBusy
                               # (SQ_ACTIVE_INST_VALU x 4 / SIMDs / GRBM_GUI_ACTIVE: the share of the SIMD cycles in which the vector ALU was executing; ~4.1 cycles per
                               # instruction in every kernel here, what tools/valu_rate.hip measures for this opcode mix).  ONE rule for every kernel:
                               # the issue-bound classes report valu_busy (measured), lane_util (active lanes per issued instruction) and their product — the share of
                               # the chip's VALU lane-slots that did useful work — as `frac`; valu_issue (the 2-cycle rule) stays beside it.  No per-kernel constant.
N_SIMD = 1024                  # 256 CUs x 4 SIMDs
KERNEL_OF_CLASS = {"trace": "k_trace", "shadow": "k_shadow", "shade": "k_shade", "media": "k_track+k_scatter", "select": "k_light_select"}
CLASSES = ("trace", "shadow", "shade", "media", "select")


def class_times(st):
    """hk_stats -> seconds and launches per kernel class.  `select` — the next-event light of every vertex chosen by a kernel of its own in
    scenes with a deep light BVH — is part of K9 and of hk_stats.seconds_shade; here it is a class of its own and `shade` is the rest."""
    sel = float(st.seconds_select)
    timed = dict(trace=st.seconds_trace, shadow=st.seconds_shadow, shade=max(st.seconds_shade - sel, 0.0), media=st.seconds_media, select=sel, other=st.seconds_other)
    launches = dict(trace=int(st.trace_launches), shadow=int(st.shadow_launches), shade=int(st.shade_launches), media=int(st.media_launches), select=int(st.select_launches))
    return timed, launches


def class_rooflines(config, timed, launches, sc, default_frame):
    """One entry per hot kernel class.  IN-RUN: seconds / avg_launch_ms (HIP events on the launch stream) and the counted units behind
    the SURVEY 8(d) algorithmic bytes.  FROM COMMITTED rocprofv3 PASSES of the same workload (named in `counters_from`): HBM traffic,
    L2 hit rate, VALU / SALU instruction counts, lane utilisation.  Shading and media classes are priced against HBM; the BVH
    traversal classes of surface scenes against instruction issue (their nodes come from LDS / L1 / L2: a byte ceiling says nothing)."""
    alg = dict(trace=int(sc.bytes_algorithmic_trace), shadow=int(sc.bytes_algorithmic_shadow), shade=int(sc.bytes_algorithmic_shade),
               media=int(sc.bytes_algorithmic_media), select=0)
    if timed.get("select", 0) > 0:      # SURVEY 8(d) charges 60 B per light-BVH node visited: those visits happen in the selection kernel
        alg["select"] = min(60 * int(sc.light_bvh_nodes), alg["shade"])
        alg["shade"] -= alg["select"]

    def committed(stem):
        rel = os.path.join("profiles", "%s_%s.json" % (stem, config))
        try:    # the PMC passes were taken on one GPU at the config's default frame
            return (json.load(open(os.path.join(ROOT, rel))), rel) if default_frame else ({}, None)
        except (OSError, ValueError):
            return {}, None

    (pmc, pmc_file), (util, util_file) = committed("pmc_traffic"), committed("utilisation")
    walk = int(sc.shadow_collisions) > 0
    rooflines = []
    for cls in CLASSES:
        n_launch = max(launches.get(cls, 0), 1)
        avg_s = timed.get(cls, 0.0) / n_launch
        if timed.get(cls, 0.0) <= 0:
            continue
        kernel = KERNEL_OF_CLASS[cls]
        names = ["k_track", "k_scatter"] if cls == "media" else [kernel]
        if cls == "shadow" and walk and ("k_walk" in pmc or "k_walk" in util):
            names = ["k_walk"]      # the pooled shadow walk of a grey medium (k_walk_pool) is its own kernel family in the profiles
        tr = [pmc.get(k, {}) for k in names]
        u = util.get(names[0], {})
        bytes_rate = alg[cls] / n_launch / avg_s / 1e9
        e = {"kernel": kernel}
        # instruction issue, re-priced on this run's launch time: the committed pass gives the wave-level VALU instruction count per launch
        # (a property of the workload), the clock (its own cycles / its own duration) and the lane utilisation
        issue = busy = None
        if u.get("valu_inst_per_launch") and u.get("valu_issue_frac") and u.get("avg_launch_us"):
            clock_hz = u["valu_inst_per_launch"] * VALU_CYCLES / (u["valu_issue_frac"] * N_SIMD) / (u["avg_launch_us"] * 1e-6)
            issue = u["valu_inst_per_launch"] * VALU_CYCLES / (avg_s * clock_hz * N_SIMD)
        if u.get("valu_busy") and u.get("avg_launch_us"):
            busy = min(u["valu_busy"] * (u["avg_launch_us"] * 1e-6) / avg_s, 1.0)     # the same busy cycles per launch over THIS run's launch time
        useful = (busy if busy is not None else issue)
        useful = useful * u["lane_util"] if (useful is not None and u.get("lane_util")) else None
        # the light selection walks a tree whose nodes come from LDS / L2 like the traversal kernels' (the 60 B per node of SURVEY 8d are an
        # upper bound of its HBM need): priced against instruction issue, as they are
        traversal = cls == "select" or (cls in ("trace", "shadow") and not (cls == "shadow" and walk))
        if traversal and useful is None:
            # no committed counter pass for this workload: the ceiling that binds a traversal kernel cannot be priced from this run alone
            e.update({"bound": "valu_issue", "achieved": None, "peak": 1.0, "unit": "share of the chip's VALU lane-slots doing useful work", "frac": None,
                      "algorithmic_bytes_per_launch": int(alg[cls] / n_launch), "algorithmic_gbs": round(bytes_rate, 2)})
        elif traversal:
            e.update({"bound": "valu_issue", "achieved": round(useful, 4), "peak": 1.0, "unit": "share of the chip's VALU lane-slots doing useful work", "frac": round(useful, 4),
                      "valu_busy_in_run": round(busy, 4) if busy is not None else None, "valu_issue_in_run": round(issue, 4) if issue is not None else None,
                      "valu_instructions_per_launch": u["valu_inst_per_launch"],
                      "algorithmic_bytes_per_launch": int(alg[cls] / n_launch), "algorithmic_gbs": round(bytes_rate, 2)})
        else:
            e.update({"bound": "hbm", "achieved": round(bytes_rate, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(bytes_rate / HBM_PEAK_GBS, 5),
                      "algorithmic_bytes_per_launch": int(alg[cls] / n_launch)})
            if useful is not None:
                e["useful_lane_issue"] = round(useful, 4)     # valu_busy (in run) x lane_util
        e.update({"avg_launch_ms": round(avg_s * 1e3, 4), "launches": n_launch, "seconds": round(timed[cls], 4)})
        if all("hbm_bytes_per_launch" in t for t in tr):
            # the PMC launches average over the same depths as the bench's (whole passes): traffic per launch is comparable
            e["traffic"] = sum(t["hbm_bytes_per_launch"] * t.get("launches", 1) for t in tr) / max(sum(t.get("launches", 1) for t in tr), 1)
            e["traffic_over_algorithmic"] = round(e["traffic"] / max(alg[cls] / n_launch, 1), 3)
            e["hbm_frac_by_traffic"] = round(e["traffic"] / avg_s / 1e9 / HBM_PEAK_GBS, 4)
        else:
            e["traffic"] = None
        if "l2_hit_rate" in tr[0]:
            e["l2_hit_rate"] = tr[0]["l2_hit_rate"]
        for k in ("valu_busy", "valu_issue_frac", "lane_util", "wait_frac", "waves_per_simd"):
            if k in u:
                e[k] = u[k]
        if u.get("salu_per_launch") and u.get("valu_inst_per_launch"):
            e["salu_over_valu"] = round(u["salu_per_launch"] / u["valu_inst_per_launch"], 3)
        e["counters_from"] = [f for f in (pmc_file if e["traffic"] is not None else None, util_file if u else None) if f]
        e["measured_in_run"] = ["avg_launch_ms", "seconds", "launches", "algorithmic_bytes_per_launch"]
        hb, vi, vb = e.get("hbm_frac_by_traffic"), e.get("valu_issue_frac"), e.get("valu_busy")
        if hb is None or vi is None:
            e["binding"] = "unprofiled on this workload"
        elif hb >= 0.5 and (vb is None or hb >= vb):
            e["binding"] = "HBM traffic (%.0f %% of the 8 TB/s peak, %.0f %% of the device-copy rate); VALU pipes busy %.0f %% of the SIMD cycles" % (
                100 * hb, 100 * hb * HBM_PEAK_GBS / 5200.0, 100 * (vb if vb is not None else vi))
        elif vb is not None:
            e["binding"] = ("the vector ALU: busy %.0f %% of the SIMD cycles (SQ_ACTIVE_INST_VALU; %.0f %% by the 2-cycles-per-instruction rule) with %.0f %% of the lanes active "
                            "= %.0f %% useful lane-slots, at %.1f resident waves per SIMD; HBM traffic %.0f %% of peak" % (
                                100 * vb, 100 * vi, 100 * e.get("lane_util", 0), 100 * vb * e.get("lane_util", 0), e.get("waves_per_simd", 0), 100 * hb))
        else:
            e["binding"] = "instruction issue and latency: VALU issue %.0f %% of the SIMD cycles (2 cycles per instruction) with %.0f %% of the lanes active = %.0f %% useful lane-slots; HBM traffic %.0f %% of peak" % (
                100 * vi, 100 * e.get("lane_util", 0), 100 * vi * e.get("lane_util", 0), 100 * hb)
        # BVH nodes come from LDS / L1 / L2, not HBM: the SURVEY 8(d) byte formula is an upper bound of traversal's HBM need, not a ceiling
        if traversal:
            e["note"] = "algorithmic bytes count every BVH node / triangle visit at full size (SURVEY 8d); nodes are served from LDS / L1 / L2, measured HBM traffic is the `traffic` field"
        if cls == "media":
            e["units"] = {"collisions": int(sc.track_collisions), "dda_steps": int(sc.track_dda_steps), "scatter_vertices": int(sc.scatter_vertices),
                          "bytes_per_collision": {"nanovdb_scene": 84, "grid_scene": 36}, "bytes_per_dda_step": 4}   # (hk_stats prices each scene's collisions by its own media)
        if cls == "shadow" and walk:
            e["units"] = {"collisions": int(sc.shadow_collisions), "dda_steps": int(sc.shadow_dda_steps), "casts": int(sc.rays_shadow)}
        rooflines.append(e)
    return rooflines


def frame_check_verdict(fc, media):
    """The pass / fail rule of `frame_check` (bench.py exits with code 4 on a failure).  Surfaces: SURVEY 8(d)'s frame tolerance on the same sample
    indices (relMSE <= 1e-3, >= 99 % of the pixels within 1e-2) and the timed film's mean within 2 % of the oracle's.  Media scenes: both means within
    2 % and >= 85 % of the pixels within 1e-2 sample for sample (their paths decorrelate at the first ulp; the converged bound is tests/test_converged_parity.py)."""
    if not fc.get("finite") or abs(fc.get("mean_ratio", 0.0) - 1.0) > 0.02:
        return False
    if media:
        return bool(abs(fc.get("same_samples_mean_ratio", 0.0) - 1.0) <= 0.02 and fc.get("same_samples_frac_pixels_within_1e-2", 0.0) >= 0.85)
    return bool(fc.get("same_samples_rel_mse", 1.0) <= 1e-3 and fc.get("same_samples_frac_pixels_within_1e-2", 0.0) >= 0.99)


LINE_LIMIT = 3500              # the driver keeps the last ~8 000 characters of stdout: the ONE JSON line stays well inside them


def _clip(text, n):
    text = str(text)
    return text if len(text) <= n else text[:n - 3] + "..."


def _pick(d, keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d and d[k] is not None}


def compact_line(result, limit=LINE_LIMIT, detail_file="bench_detail.json"):
    """The ONE stdout line: the contract's headline fields + `roofline` + `cpu_baseline` + one short entry per other BASELINE.json config
    + the one-sample-per-call figures, at most `limit` characters.  Everything else the run measured (per-class `rooflines`, unit counts,
    notes) is in `detail_file` and on stderr."""
    head = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")
    line = {k: result.get(k) for k in head}
    cfg = dict(result.get("config") or {})
    cfg["workload"] = _clip(cfg.get("workload", ""), 200)
    if "parallelism" in cfg:
        cfg["parallelism"] = _clip(cfg["parallelism"], 90)
    line["config"] = cfg
    for k in ("seconds_per_frame", "seconds_to_256spp", "cold_frame_seconds", "rays"):
        if k in result:
            line[k] = result[k]
    r = result.get("roofline") or {}
    line["roofline"] = {k: r.get(k) for k in ("bound", "achieved", "peak", "unit", "frac", "traffic")}
    line["roofline"].update(_pick(r, ("kernel", "avg_launch_ms", "launches", "algorithmic_bytes_per_launch", "hbm_frac_by_traffic",
                                      "traffic_over_algorithmic", "traffic_frac_of_measured_copy", "valu_busy", "lane_util", "measured_copy_gbs", "kernel_seconds")))
    cpu = result.get("cpu_baseline")
    line["cpu_baseline"] = None if cpu is None else dict(_pick(cpu, ("value", "unit", "cores", "kind", "seconds_per_frame_extrapolated")),
                                                           sample=_clip(cpu.get("sample", ""), 110))
    if isinstance(result.get("frame_check"), dict):
        line["frame_check"] = {k: v for k, v in result["frame_check"].items() if k != "tolerance"}
    if result.get("configs") is not None:
        line["configs"] = []
        for c in result["configs"]:
            e = _pick(c, ("config", "seconds_per_frame", "value", "unit", "frames_timed", "warmup_frames", "error"))
            if isinstance(c.get("frame_check"), dict):
                e["frame_ok"] = bool(c["frame_check"].get("ok"))
            if "workload" in c:
                e["workload"] = _clip(c["workload"], 120)
            if isinstance(c.get("roofline"), dict):
                e["roofline"] = _pick(c["roofline"], ("kernel", "bound", "frac", "hbm_frac_by_traffic", "valu_busy", "lane_util", "avg_launch_ms"))
            if isinstance(c.get("kernel_seconds"), dict):
                e["kernel_seconds"] = c["kernel_seconds"]
            if isinstance(c.get("progressive"), dict):
                e["progressive"] = _pick(c["progressive"], ("ms_per_call", "ms_per_call_with_readback", "ms_per_call_pipelined_readback", "vs_frame_sample"))
            line["configs"].append(e)
    if isinstance(result.get("progressive"), dict):
        line["progressive"] = _pick(result["progressive"], ("calls", "one_launch_per_call", "ms_per_call", "ms_per_call_with_readback", "ms_per_call_pipelined_readback", "ms_per_call_batched_no_readback",
                                                            "ms_per_sample_of_the_full_frame", "vs_frame_sample", "vs_frame_sample_with_readback", "error"))
    line["detail"] = detail_file
    # whatever a future field adds, the line never outgrows the driver: shed the least important parts first
    shed = [lambda: line.pop("rays", None),
            lambda: [c.pop("kernel_seconds", None) for c in line.get("configs", [])],
            lambda: [c.pop("progressive", None) for c in line.get("configs", [])],
            lambda: line["roofline"].pop("kernel_seconds", None),
            lambda: [c.pop("workload", None) for c in line.get("configs", [])],
            lambda: line["config"].update(workload=_clip(line["config"]["workload"], 100)),
            lambda: line.pop("configs", None),
            lambda: line.pop("progressive", None)]
    for drop in shed:
        if len(json.dumps(line)) <= limit:
            break
        drop()
    return line


def emit(result, detail_path=None):
    """Full record -> bench_detail.json (working directory) and stderr; the compact line -> stdout, the only stdout line of the run."""
    detail_path = detail_path or os.path.join(os.getcwd(), "bench_detail.json")
    full = json.dumps(result)
    try:
        with open(detail_path, "w") as f:
            f.write(full + "\n")
    except OSError as e:
        sys.stderr.write("bench.py: could not write %s (%s)\n" % (detail_path, e))
    sys.stderr.write("bench.py detail: " + full + "\n")
    sys.stderr.flush()
    text = json.dumps(compact_line(result, detail_file=os.path.basename(detail_path)))
    assert len(text) <= LINE_LIMIT, len(text)
    print(text)
    sys.stdout.flush()


def progressive_line(vp, scene, film, cam, calls, frame_spp, seconds_per_frame, torch):
    """The reference's interactive path: render!(vp, scene, film, camera) = ONE sample of every pixel per call (volpath.jl:445-450, 471-474),
    here hk_render(first = i, n = 1) for i = 1 .. calls on a cleared film, (a) back to back and (b) each followed by hk_film_read_rgb (the
    frame an interactive viewer shows: finalize kernel + the device-to-host copy of the RGB frame).  `vs_frame_sample` = time per call over
    the time per sample of the config's full frame (the 256-spp pass has ~100 paths per resident lane in flight, a 1-spp call has < 1)."""
    submit = [0.0]
    fused = [0]

    def run(readback):
        vp.clear()
        film.iteration_index = 0
        vp.reset_stats()
        vp.sync()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(calls):
            vp.render_samples(scene, film, cam, 1, first=i + 1, readback=readback)
        submit[0] = time.perf_counter() - t0
        if readback == "pipelined":
            vp.finish_pipelined(film)            # the last call's frame
        vp.sync()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        st = vp.stats()
        fused[0] = int(st.fused_passes)
        return dt, int(st.rays_closest) + int(st.rays_shadow)

    run(False)                       # untimed: the one-sample pass's path state / tables
    dt, rays = run(False)            # the library's default: small calls that continue each other are noted and rendered as one pass (hk_render_tile)
    host_ms = submit[0] / calls * 1e3
    with vp._ctx.options(HK_BATCH_PATHS_M=0):    # every call rendered at once
        run(False)
        dt_each = run(False)[0]
        one_launch = fused[0] == calls        # k_small_pass: camera rays, every bounce and the film update of a call in ONE launch (hk_stats.fused_passes)
    # a read-back after every call (nothing to batch): "view" = hk_film_read_rgb into ONE host buffer, which the library pins after the second
    # call; "pipelined" = hk_film_read_rgb_async / hk_film_read_wait, the frame shown lags one call and the GPU never waits for the host
    dt_rb = dt_pipe = float("nan")
    if READBACK_PASS:
        run("view")
        dt_rb = run("view")[0]
        run("pipelined")
        dt_pipe = run("pipelined")[0]
    per_sample = seconds_per_frame / max(frame_spp, 1)
    # `ms_per_call` is what ONE render! call costs a caller that wants its sample rendered now (batching off, no read-back);
    # `..._with_readback` adds the frame an interactive viewer shows after every call; the batched figure is a 64-spp pass in disguise
    return {"calls": calls, "one_launch_per_call": bool(one_launch), "ms_per_call": round(dt_each / calls * 1e3, 4), "ms_per_call_with_readback": round(dt_rb / calls * 1e3, 4),
            "ms_per_call_pipelined_readback": round(dt_pipe / calls * 1e3, 4),
            "ms_per_call_batched_no_readback": round(dt / calls * 1e3, 4), "value_batched": round(rays / dt / 1e6, 2), "unit": "Mrays/s",
            "host_ms_per_call_batched": round(host_ms, 4), "ms_per_sample_of_the_full_frame": round(per_sample * 1e3, 4),
            "vs_frame_sample": round(dt_each / calls / per_sample, 3), "vs_frame_sample_with_readback": round(dt_rb / calls / per_sample, 3),
            "vs_frame_sample_pipelined_readback": round(dt_pipe / calls / per_sample, 3), "vs_frame_sample_batched": round(dt / calls / per_sample, 3),
            "note": "calls without a read-back in between are batched by the library (HK_BATCH_PATHS_M); `ms_per_call` is the loop with batching off; "
                    "with_readback = hk_film_read_rgb after every call, pipelined = hk_film_read_rgb_async + hk_film_read_wait (frame shown one call late)"}


def one_frame_line(hk, scenes, torch, config, device):
    """ONE warm frame of another BASELINE.json config on this GPU, after the bench line's timed region: wall-clock seconds per frame,
    Mrays/s and the per-class rooflines (same definitions as the bench line).  Four frames are rendered: a first one that uploads the
    scene, builds the sampler tables and counts the units, a second untimed one, the timed one, and a replay with HIP events around
    every launch."""
    import numpy as np
    t_setup = time.perf_counter()
    free_b, total_b = torch.cuda.mem_get_info()
    scene, film, cam, W, H, depth, spp, workload = build_workload(config, scenes)
    accum = torch.zeros(4 * W * H, dtype=torch.float32, device="cuda")
    vp = hk.VolPath(max_depth=depth, samples=max(spp, 256), device=device)
    vp.use_external_accumulators(accum.data_ptr())
    vp._ensure(film)

    def frame():
        vp.clear()
        vp.render_samples(scene, film, cam, spp, stride=1, first=1, readback=False)
        vp.sync()
        torch.cuda.synchronize()

    vp.enable_counters(count_nodes=True, time_kernels=False)
    vp.reset_stats()             # (the statistics belong to the device context, which the bench line's frames have used)
    frame()
    sc = vp.stats()
    setup_s = time.perf_counter() - t_setup
    vp.enable_counters(count_nodes=False, time_kernels=False)
    frame()                      # (untimed: the first launch of the kernel instantiations that do not count)
    vp.reset_stats()
    t0 = time.perf_counter()
    frame()
    seconds = time.perf_counter() - t0
    st = vp.stats()
    rays = int(st.rays_closest) + int(st.rays_shadow)
    acc_np = accum.detach().cpu().numpy()
    frame_check = {"finite": bool(np.isfinite(acc_np).all() and (acc_np[:3 * W * H] >= 0).all()), "mean_rgb_sum": round(float(acc_np[:3 * W * H].mean()), 6)}
    frame_check["ok"] = frame_check["finite"] and frame_check["mean_rgb_sum"] > 0
    vp.enable_counters(count_nodes=False, time_kernels=True)
    vp.reset_stats()
    frame()
    tk = vp.stats()
    timed, launches = class_times(tk)
    rooflines = class_rooflines(config, timed, launches, sc, True)      # (no committed counter passes for the two-spheres variant: in-run fields only)
    dom = max((k for k in CLASSES if timed[k] > 0), key=lambda k: timed[k])
    line = {"config": config, "workload": "%s, %d spp per frame" % (workload, spp), "resolution": [W, H], "max_depth": depth, "spp_per_frame": spp,
            "triangles": int(scene.desc.n_triangles), "lights": int(scene.desc.n_lights), "frames_timed": 1, "free_hbm_gb_before": round(free_b / 1e9, 1),
            "warmup_frames": 2, "seconds_per_frame": round(seconds, 4), "value": round(rays / seconds / 1e6, 2), "unit": "Mrays/s",
            "rays": {"closest": int(st.rays_closest), "shadow": int(st.rays_shadow), "medium_collisions": int(st.medium_collisions)},
            "kernel_seconds": {k: round(v, 4) for k, v in timed.items()}, "setup_seconds": round(setup_s, 2),
            "roofline": next(e for e in rooflines if e["kernel"] == KERNEL_OF_CLASS[dom]), "rooflines": rooflines, "frame_check": frame_check}
    vp.enable_counters(count_nodes=False, time_kernels=False)
    vp.close()
    del accum
    torch.cuda.empty_cache()
    # what the one-sample-per-call measurement needs later (ALL full frames of the run are rendered first: bench.py docstring)
    line["_progressive_inputs"] = (scene, film, cam, depth, spp, seconds)
    return line


def progressive_of(hk, torch, device, scene, film, cam, depth, spp, seconds, calls):
    """progressive_line on a fresh integrator (the scene is still resident)"""
    vp = hk.VolPath(max_depth=depth, samples=max(spp, 256), device=device)
    vp._ensure(film)
    try:
        return progressive_line(vp, scene, film, cam, calls, spp, seconds, torch)
    finally:
        vp.close()


def child_config_line(config, progressive_calls):
    """`bench.py --config <config>` in a child process (one warm frame timed, its class times, rooflines and one-sample-per-call path):
    the fields of its JSON line that a `configs` entry carries."""
    import tempfile
    fd, detail = tempfile.mkstemp(prefix="hk_bench_%s_" % config, suffix=".json")
    os.close(fd)
    cmd = [sys.executable, os.path.abspath(__file__), "--config", config, "--steps", "1", "--warmup", "2", "--no-cpu-baseline",
           "--progressive", str(progressive_calls), "--detail-file", detail]
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE"):
        env.pop(k, None)
    try:
        out = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
        if out.returncode != 0:
            raise RuntimeError("exit code %d: %s" % (out.returncode, out.stderr.decode(errors="replace")[-300:]))
        with open(detail) as f:
            d = json.load(f)
    finally:
        try:
            os.unlink(detail)
        except OSError:
            pass
    cfg = d["config"]
    return {"config": config, "workload": cfg["workload"], "resolution": cfg["resolution"], "max_depth": cfg["max_depth"], "spp_per_frame": cfg["spp_per_frame"],
            "triangles": cfg["triangles"], "lights": cfg["lights"], "frames_timed": d["steps"], "warmup_frames": d["warmup"], "seconds_per_frame": d["seconds_per_frame"],
            "cold_frame_seconds": d["cold_frame_seconds"], "value": d["value"], "unit": d["unit"], "rays": d["rays"],
            "kernel_seconds": d["roofline"]["kernel_seconds"], "setup_seconds": d["setup_seconds"], "roofline": d["roofline"], "rooflines": d["rooflines"],
            "progressive": d.get("progressive"), "frame_check": d.get("frame_check"), "measured_in": "a process of its own (bench.py --config %s --steps 1 --warmup 2), before this process touched the GPU" % config}


def main():
    args = parse_args()
    world_env = os.environ.get("WORLD_SIZE")
    if args.gpus > 1 and world_env is None:
        launch_ranks(args)          # never returns

    # The other north_star targets (`configs` of the line) are measured FIRST, each by `bench.py --config X` in a child process, while this
    # process has not touched the GPU yet.  Measured AFTER this process's own frames, timed replays and copies — the parent alive with
    # whatever queues those left it — the children's cloud frame came out at 0.680 s instead of 0.629 s and the many-light frame at
    # 1.54 s instead of 1.43 s: the per-launch cost of a second hardware queue (DESIGN.md §5 "two speeds") does not stop at the process
    # boundary (a process that only holds an idle context has no such effect).  Children that fail are measured in this process after
    # the bench line's own frames (`measured_in` says which).
    pre_configs = {}
    if args.extra_configs and int(world_env or "1") == 1:
        for c in os.environ.get("HK_BENCH_EXTRAS", "cornell_sphere_box,cloud,sky,manylight").split(","):
            try:
                pre_configs[c] = child_config_line(c, min(args.progressive, 32))
            except Exception as e:               # noqa: BLE001
                sys.stderr.write("bench.py: %s in its own process failed (%s: %s); it will be measured in this process\n" % (c, type(e).__name__, e))

    import numpy as np              # noqa: F401
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(world_env or "1")
    # HK_BENCH_SINGLE_DEVICE=1 (test hook): every rank shares cuda:0 and the process group is gloo, so the N > 1 code path can be
    # exercised on a 1-GPU box.  The driver's runs never set it: one process per GPU over RCCL ("nccl").
    single_device = os.environ.get("HK_BENCH_SINGLE_DEVICE") == "1"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group("gloo" if single_device else "nccl", rank=rank, world_size=world)
    # Host-side coordination (barriers, flags, the final statistics, the 128-byte communicator id) travels over gloo on CPU tensors, never
    # over the NCCL process group: its first collective would give this process a stream — a hardware queue — of its own, and from then
    # on every kernel launch of the frames costs 50 - 120 us more (DESIGN.md §5 "two speeds").  torch.distributed's RCCL reduce is only
    # used for the cross-check of the in-library reduce, after the timed region (or as the fallback when the in-library one is unavailable).
    coord, coord_on_device = None, False
    if world > 1 and not single_device:
        try:
            coord = dist.new_group(backend="gloo")
        except Exception as e:                   # noqa: BLE001  (no gloo transport on this node: coordinate over the NCCL group after all)
            sys.stderr.write("bench.py: no gloo group (%s: %s); coordinating over the NCCL process group\n" % (type(e).__name__, e))
            coord, coord_on_device = None, True

    def host_all_reduce(values, op):
        t = torch.tensor(values, dtype=torch.float64, device="cuda" if coord_on_device else "cpu")
        if world > 1:
            dist.all_reduce(t, op=op, group=coord)
        return [float(x) for x in t.cpu()]
    if single_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)

    import hikari_jl_amd as hk
    from hikari_jl_amd import distributed as hd
    from hikari_jl_amd import scenes

    scene, film, cam, W, H, DEPTH, FULL_SPP, workload = build_workload(args.config, scenes)
    if args.spp:
        FULL_SPP = args.spp
    if args.steps is None:
        args.steps = 20 if (args.config.startswith("cornell") and FULL_SPP <= 256) else 1
    strong = args.scaling == "strong"
    # this rank's share of a frame: sample indices first, first + stride, ... (count of them)
    first, my_spp, stride = hd.shard_samples(FULL_SPP, rank, world) if strong else (rank + 1, FULL_SPP, world)
    frame_spp = FULL_SPP if strong else FULL_SPP * world
    if single_device and args.spp_per_pass == 0:
        args.spp_per_pass = max(1, min(my_spp, 64))     # several contexts share one device: bounded path state per rank
    n_pix = W * H
    # film accumulators live in a torch tensor so torch.distributed (RCCL) can reduce them in place (the cross-check)
    accum = torch.zeros(4 * n_pix, dtype=torch.float32, device="cuda")
    # `samples` only sizes the ZSobol index (log2 of max(samples, 4096)): cover every sample index this run touches
    vp = hk.VolPath(max_depth=DEPTH, samples=max(FULL_SPP * (1 if strong else world), 256), samples_per_pass=args.spp_per_pass, device=local_rank)
    vp.use_external_accumulators(accum.data_ptr())
    vp._ensure(film)

    def barrier():
        if world > 1:
            dist.barrier(group=coord)
        vp.sync()
        torch.cuda.synchronize()

    # ---- the film reduce of N > 1: hk_comm over RCCL inside the library (the C-ABI's own exchange step) ----
    comm = None
    comm_note = None
    if world > 1 and not single_device:
        # If the in-library communicator cannot be set up on every rank (or its first reduce fails, below), ALL ranks fall back to
        # torch.distributed's reduce — the same RCCL ring on the same buffer — and the line says so.  The bring-up calls are
        # collective: each runs under a timeout, and a rank that does not come back ends the run with a non-zero exit.
        ok = 1
        try:
            uid = [hk.Comm.unique_id() if rank == 0 else None]
        except Exception as e:           # noqa: BLE001
            uid, ok, comm_note = [None], 0, "hk_comm_unique_id: %s" % e
        dist.broadcast_object_list(uid, src=0, group=coord)        # the launcher's side channel for the 128-byte id
        if uid[0] is None:
            ok = 0
        if ok:
            try:
                comm = call_with_timeout(lambda: hk.Comm.rank(vp._ctx, uid[0], rank, world), COMM_TIMEOUT_S, "hk_comm_create_rank")
            except Exception as e:       # noqa: BLE001
                ok, comm_note = 0, "hk_comm_create_rank: %s" % e
        if int(host_all_reduce([ok], dist.ReduceOp.MIN)[0]) == 0:
            comm = None
            comm_note = comm_note or "another rank could not create its communicator"

    def reduce_films():
        if world == 1:
            return
        if comm is not None:
            comm.reduce_films([vp], root=0)
        else:                                         # fallback / HK_BENCH_SINGLE_DEVICE test hook (gloo stages through the host)
            vp.sync()
            hd.reduce_film(accum, root=0)

    def run_frames(n_frames, first_idx=first, reduce=True):
        # One frame: clear the film, this rank's samples of every pixel, the reduce onto rank 0.  Every frame renders the same
        # sample indices (a frame loop), so the sampler tables built by the first frame serve the later ones.
        for _ in range(n_frames):
            vp.clear()
            if my_spp > 0:
                vp.render_samples(scene, film, cam, my_spp, stride=stride, first=first_idx, readback=False)
            if reduce:
                reduce_films()

    # ---- warmup (untimed): also uploads the scene / builds the BVH / allocates the path state ----
    t0 = time.time()
    run_frames(max(args.warmup, 1), reduce=False)
    barrier()
    setup_s = time.time() - t0
    reduce_check = None
    if world > 1:
        # untimed: RCCL sets up the reduce's channels on first use.  The warm-up film before and after the in-library reduce is kept
        # for the cross-check against torch.distributed's reduce (same inputs, same ring: identical sums), which runs AFTER the timed
        # region (see `coord` above)
        keep = accum.clone()
        if comm is not None:
            ok = 1
            try:
                call_with_timeout(lambda: (reduce_films(), vp.sync()), COMM_TIMEOUT_S, "the first hk_film_reduce")
            except Exception as e:       # noqa: BLE001
                ok, comm_note = 0, "hk_film_reduce: %s" % e
            if int(host_all_reduce([ok], dist.ReduceOp.MIN)[0]) == 0:
                comm = None
                comm_note = comm_note or "hk_film_reduce failed on another rank"
                accum.copy_(keep)
                reduce_films()
        else:
            reduce_films()
        barrier()
        mine = accum.clone()
    # ---- a cold frame (untimed for `value`): the sample-bit table of a NEW sample range is built inside it, as a one-shot
    #      `integrator(scene, film, camera)` call pays it; path state and scene stay resident ----
    shift = stride * ((FULL_SPP + stride - 1) // stride + 16)
    barrier()
    t0 = time.perf_counter()
    run_frames(1, first_idx=first + shift)
    barrier()
    cold_s = time.perf_counter() - t0
    run_frames(1, reduce=False)              # back to the bench's sample range (rebuilds its table, untimed)
    accum.zero_()
    vp.enable_counters(count_nodes=False, time_kernels=False)
    vp.reset_stats()
    barrier()

    # ---- timed region: EXACTLY K steps (frames), each with its reduce ----
    t0 = time.perf_counter()
    run_frames(args.steps)
    barrier()
    elapsed = time.perf_counter() - t0
    st = vp.stats()
    rays_local = int(st.rays_closest) + int(st.rays_shadow)
    elapsed_max, cold_max = host_all_reduce([elapsed, cold_s], dist.ReduceOp.MAX) if world > 1 else (elapsed, cold_s)
    total_rays = host_all_reduce([float(rays_local)], dist.ReduceOp.SUM)[0] if world > 1 else float(rays_local)
    if world > 1:
        # the cross-check of the in-library reduce (untimed, after the timed region): torch.distributed's reduce of the same warm-up film
        last = accum.clone()
        accum.copy_(keep)
        hd.reduce_film(accum, root=0)
        barrier()
        if rank == 0:
            reduce_check = bool(torch.allclose(mine, accum, rtol=1e-6, atol=1e-7))
        accum.copy_(last)           # (the film of the timed frames, as the replays below expect it)
        del keep, mine, last

    result = None
    if rank == 0:
        # ---- per-kernel-class times: an untimed replay of the same frames (this rank's share) with HIP events around every launch,
        #      on the launch stream.  With the events on, the library keeps everything on one stream (the timed region above runs the
        #      shadow rays of bounce d beside the traversal of bounce d + 1 on a second stream), so the class times add up to slightly
        #      MORE than the timed frame ----
        accum_timed = accum.clone()
        vp.enable_counters(count_nodes=False, time_kernels=True)
        vp.reset_stats()
        run_frames(args.steps, reduce=False)
        vp.sync()
        st = vp.stats()
        timed, launches = class_times(st)
        # ---- counts from an instrumented (untimed) replay of the same frames ----
        vp.enable_counters(count_nodes=True, time_kernels=False)
        vp.reset_stats()
        run_frames(args.steps, reduce=False)
        vp.sync()
        sc = vp.stats()
        accum.copy_(accum_timed)
        del accum_timed
        vp.enable_counters(False, False)
        # ---- per-kernel-class ceilings: ALGORITHMIC bytes from the counted replay (hk_stats.bytes_algorithmic_*) over the
        #      HIP-event time of that class's launches; PMC traffic / L2 hit rate / VALU issue / lane utilisation from the rocprofv3
        #      passes committed under profiles/ for the same workload (tools/profile_round.sh) ----
        default_frame = world == 1 and not args.spp and not args.spp_per_pass
        rooflines = class_rooflines(args.config, timed, launches, sc, default_frame)
        dom = max((k for k in CLASSES if timed[k] > 0), key=lambda k: timed[k])
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
        roofline = dict(next(e for e in rooflines if e["kernel"] == KERNEL_OF_CLASS[dom]))
        roofline.update({"measured_copy_gbs": round(copy_gbs, 1), "traffic_frac_of_measured_copy": round(roofline["hbm_frac_by_traffic"] * HBM_PEAK_GBS / copy_gbs, 4) if roofline.get("hbm_frac_by_traffic") else None,   # (measured bytes over the measured ceiling: never above 1; `frac` prices SURVEY 8d's algorithmic bytes, which exceed what the compact records move)
                         "nodes_per_cast": round(int(sc.trace_nodes) / max(int(sc.rays_closest), 1), 2),
                         "tris_per_cast": round(int(sc.trace_tris) / max(int(sc.rays_closest), 1), 2),
                         "kernel_seconds": {k: round(v, 4) for k, v in timed.items()}})

        # ---- CPU baseline: the oracle (a port, NOT Julia / KernelAbstractions.CPU()) on a bounded sample ----
        cpu = None
        if not args.no_cpu_baseline and world == 1:
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
            oracle_acc, ost = osc.render(p, cam, W, H, args.cpu_spp)       # (kept: the timed film is checked against this image below)
            cdt = time.perf_counter() - c0
            crays = int(ost.rays_closest) + int(ost.rays_shadow)
            cpu = {"value": round(crays / cdt / 1e6, 4), "unit": "Mrays/s", "cores": oracle.max_threads(), "kind": "port",
                   "sample": "%d spp of the same %dx%d depth-%d frame (%.1f s); CPU restatement of Hikari VolPath, not Julia" % (args.cpu_spp, W, H, DEPTH, cdt),
                   "seconds_per_frame_extrapolated": round(cdt * FULL_SPP / args.cpu_spp, 1)}
            osc.close()

        # ---- the film the timed region rendered is LOOKED AT: finite, and against the oracle image of the CPU leg —
        #      (a) the timed frame itself (FULL_SPP samples) vs the oracle's cpu_spp samples of the same frame: two estimates of one image,
        #      relMSE ~ var / cpu_spp + var / FULL_SPP, mean ratio within 2 %; (b) the SAME cpu_spp sample indices rendered once more on the
        #      GPU (untimed, a few ms) vs the oracle under SURVEY 8(d)'s frame tolerance: relMSE <= 1e-3 and >= 99 % of the pixels within
        #      1e-2 relative L2.  Anything else ends the run with a non-zero exit code AFTER the line is printed.
        timed_acc = accum.detach().cpu().numpy().copy()
        n_px = W * H
        frame_check = {"finite": bool(np.isfinite(timed_acc).all() and (timed_acc[:3 * n_px] >= 0).all()),
                       "mean_weight_per_pixel": round(float(timed_acc[3 * n_px:].mean()), 4), "mean_rgb_sum": round(float(timed_acc[:3 * n_px].mean()), 6)}
        frame_check["ok"] = frame_check["finite"] and frame_check["mean_rgb_sum"] > 0
        if cpu is not None and world == 1:
            def rel_mse(a, b):
                return float(np.mean((a - b) ** 2 / (b ** 2 + 1e-3)))
            ref_img = oracle.finalize(oracle_acc, W, H)
            img = oracle.finalize(timed_acc, W, H)                      # (K13 on the host: rgb / weight; the checker's finalize is three divisions per pixel)
            keep = accum.clone()
            vp.clear()
            vp.render_samples(scene, film, cam, args.cpu_spp, stride=1, first=1, readback=False)
            vp.sync()
            torch.cuda.synchronize()
            same = oracle.finalize(accum.detach().cpu().numpy().copy(), W, H)
            accum.copy_(keep)
            del keep
            d = np.sqrt(((same - ref_img) ** 2).sum(axis=2)) / np.maximum(np.sqrt((ref_img ** 2).sum(axis=2)), 1e-6)
            # A path through a medium takes hundreds of float32 tracking steps and every one feeds a comparison: where device and host
            # differ by an ulp (the documented bounds of tests/golden/ulp_bounds.json) a path takes another way and the SAME sample index
            # gives another — equally valid — estimate.  SURVEY 8(d) bounds media scenes through converged estimates
            # (tests/test_converged_parity.py); here: most pixels still agree sample for sample, and the means agree.
            media = int(st.medium_collisions) > 0
            frame_check.update({
                "oracle_spp": args.cpu_spp,
                "rel_mse_vs_oracle_%dspp" % args.cpu_spp: round(rel_mse(img, ref_img), 6),
                "mean_ratio": round(float(img.mean() / max(ref_img.mean(), 1e-12)), 5),
                "same_samples_mean_ratio": round(float(same.mean() / max(ref_img.mean(), 1e-12)), 5),
                "same_samples_rel_mse": float("%.3g" % rel_mse(same, ref_img)),
                "same_samples_frac_pixels_within_1e-2": round(float((d <= 1e-2).mean()), 5),
                "tolerance": ("media scene: mean ratio within 2 % (timed film and same samples), >= 85 % of the pixels within 1e-2 sample for sample "
                              "(paths through the medium decorrelate at the first ulp; the converged bound is tests/test_converged_parity.py)") if media else
                             "SURVEY 8(d): relMSE <= 1e-3 and >= 99 % of the pixels within 1e-2 (same samples); mean ratio within 2 % (timed film)"})
            frame_check["ok"] = frame_check_verdict(frame_check, media)

        value = total_rays / elapsed_max / 1e6
        per_frame = elapsed_max / max(args.steps, 1)
        if comm is not None or world == 1:
            par = "sample-index sharding x%d + in-library RCCL film reduce (hk_film_reduce)" % world
        elif single_device:
            par = "sample-index sharding x%d + gloo film reduce on one device (HK_BENCH_SINGLE_DEVICE test hook)" % world
        else:
            par = "sample-index sharding x%d + torch.distributed film reduce (in-library communicator unavailable: %s)" % (world, comm_note)
        result = {
            "metric": "Mrays/s", "value": round(value, 2), "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(per_frame * 1e3, 4), "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s, %d spp per frame (one step = one frame%s)" % (workload, frame_spp, "" if world == 1 else ", %d spp on each of %d GPUs" % (my_spp, world)),
                       "resolution": [W, H], "max_depth": DEPTH, "spp_per_frame": frame_spp, "spp_per_rank": my_spp, "spp_per_pass": args.spp_per_pass or min(my_spp, 256),
                       "triangles": int(scene.desc.n_triangles), "lights": int(scene.desc.n_lights), "parallelism": par,
                       "reduce_matches_torch_distributed": reduce_check},
            "seconds_timed": round(elapsed_max, 4),
            "seconds_per_frame": round(per_frame, 4),
            "seconds_to_256spp": round(per_frame * 256.0 / frame_spp, 4),
            "cold_frame_seconds": round(cold_max, 4),
            "rays": {"closest": int(st.rays_closest), "shadow": int(st.rays_shadow), "total_all_ranks": int(total_rays), "medium_collisions": int(st.medium_collisions)},
            "setup_seconds": round(setup_s, 3),
            "roofline": roofline, "rooflines": rooflines, "cpu_baseline": cpu, "frame_check": frame_check,
        }
    if comm is not None:
        comm.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    vp.close()
    if result is not None:
        if args.extra_configs and world == 1:
            # the other north_star targets, driver-visible in the same line (headline fields above are untouched): measured at the top of
            # main(), each in a process of its own; one whose child failed is measured here, in this process
            del accum
            torch.cuda.empty_cache()
            result["configs"] = []
            for c in os.environ.get("HK_BENCH_EXTRAS", "cornell_sphere_box,cloud,sky,manylight").split(","):
                line = pre_configs.get(c)
                if line is None:
                    try:
                        line = one_frame_line(hk, scenes, torch, c, local_rank)
                        inp = line.pop("_progressive_inputs", None)
                        if inp is not None and args.progressive > 0:
                            line["progressive"] = progressive_of(hk, torch, local_rank, *inp, calls=min(args.progressive, 32))
                        line["measured_in"] = "the bench line's process"
                    except Exception as e:       # noqa: BLE001
                        line = {"config": c, "error": "%s: %s" % (type(e).__name__, e)}
                result["configs"].append(line)
        if world == 1 and args.progressive > 0 and not args.spp_per_pass:
            # the one-sample-per-call path of the bench scene, after every full frame this process renders
            try:
                result["progressive"] = progressive_of(hk, torch, local_rank, scene, film, cam, DEPTH, frame_spp, per_frame, args.progressive)
            except Exception as e:               # noqa: BLE001
                result["progressive"] = {"error": "%s: %s" % (type(e).__name__, e)}
        emit(result, args.detail_file)
        bad = [] if result["frame_check"].get("ok") else [args.config]
        bad += [c.get("config") for c in result.get("configs") or [] if isinstance(c.get("frame_check"), dict) and not c["frame_check"].get("ok")]
        if bad:
            sys.stderr.write("bench.py: FRAME CHECK FAILED for %s (see frame_check in the line / detail file)\n" % ", ".join(str(b) for b in bad))
            sys.exit(4)
    return result


if __name__ == "__main__":
    main()

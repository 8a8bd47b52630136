#!/usr/bin/env python3
"""The "two speeds" of the cloud frame: what a SECOND STREAM of the process costs every later kernel launch.
    tools/two_speeds.py sequence                     one process: cloud integrator A (6 x 3 frames), B right after, a small render on a lane, C, 20 s idle, D
    tools/two_speeds.py <mode>                       one process: <what comes first>, then three timed cloud frames
        plain            nothing first
        small            a 64 x 64 Cornell render as one small call ON A LANE (HK_BATCH_PATHS_M=0 HK_PIPELINE=8 HK_PIPELINE_AFTER=0)
        small_nolanes    the same call on the context's own stream
        torch_stream     a trivial torch kernel on a torch side stream
    Environment variants are applied from outside, e.g.  GPU_MAX_HW_QUEUES=1 tools/two_speeds.py small
Output of round 4: profiles/r04_two_speeds.txt."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch                                   # noqa: E402,F401
import hikari_jl_amd as hk                     # noqa: E402
from hikari_jl_amd import scenes               # noqa: E402

LANE_ENV = {"HK_BATCH_PATHS_M": "0", "HK_PIPELINE": "8", "HK_PIPELINE_AFTER": "0"}


def small_render(on_lane):
    env = LANE_ENV if on_lane else {"HK_BATCH_PATHS_M": "0", "HK_PIPELINE": "1"}
    os.environ.update(env)
    cs, cfilm, ccam = scenes.cornell_box(64, 64, light="area")
    v = hk.VolPath(max_depth=8, samples=16)
    v(cs, cfilm, ccam)
    v.close()
    for k in env:
        os.environ.pop(k, None)


def cloud_frames(s, film, cam, groups=1):
    vp = hk.VolPath(max_depth=32, samples=256)
    vp._ensure(film)

    def frame():
        vp.clear()
        vp.render_samples(s, film, cam, 256, first=1, readback=False)
        vp.sync()

    frame()
    frame()
    out = []
    for _ in range(groups):
        t = time.perf_counter()
        for _ in range(3):
            frame()
        out.append((time.perf_counter() - t) / 3 * 1e3)
    vp.close()
    return " ".join("%.1f" % x for x in out)


def main(mode):
    s, film, cam = scenes.bomex_scene(1024, 1024)
    if mode == "sequence":
        print("A, fresh process:           ", cloud_frames(s, film, cam, 6), "ms per frame", flush=True)
        print("B, right after A:           ", cloud_frames(s, film, cam, 2), flush=True)
        small_render(True)
        print("C, after a call on a lane:  ", cloud_frames(s, film, cam, 2), flush=True)
        time.sleep(20)
        print("D, after 20 s idle:         ", cloud_frames(s, film, cam, 2), flush=True)
        return
    if mode == "small":
        small_render(True)
    elif mode == "small_nolanes":
        small_render(False)
    elif mode == "torch_stream":
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            x = torch.zeros(1 << 20, device="cuda")
            x += 1
        st.synchronize()
    print("%s: %s ms per frame" % (mode, cloud_frames(s, film, cam)), flush=True)


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "sequence")

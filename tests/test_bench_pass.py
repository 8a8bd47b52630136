"""The operating point bench.py times, tested for correctness: the headline frame is ONE wavefront pass of
164 M paths (72 GB of path state, ticketed segments, second stream) — 268 M paths in two passes for the cloud and the many-light scene —
and until round 6 no test rendered a pass larger than 10 M paths.

The film does not depend on the pass size (k_film adds a pixel's samples in sample order onto the stored accumulators, and every
random decision of a path is a function of the path alone), so the same 256 samples rendered as ONE pass and as several small passes
must give the same accumulators BIT FOR BIT and the same ray / collision counters.  The small passes stay below 2^26 records, the
large one runs the 32-bit slot / byte-offset arithmetic at 1.6 - 1.9 x 10^8 records: equality is the check of that arithmetic.
Scheduling switches (second stream, shared ticket word) are covered at the same size.

Reference: `(vp::VolPath)(scene, film, camera)` volpath.jl:655-670 renders sample after sample; bench.py's step is that frame.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _frame(hk, scene, film, cam, depth, spp, per_pass=0, options=None):
    """-> (accumulators, (closest casts, shadow casts, medium collisions)) of one `spp`-sample frame rendered `per_pass` samples at a time
    (0: the library's own pass size — what bench.py times)."""
    vp = hk.VolPath(max_depth=depth, samples=spp, samples_per_pass=per_pass)
    ctx = hk.Context.get(0)
    try:
        if options:
            with ctx.options(**options):
                vp(scene, film, cam)
                acc = vp.read_accumulators(film).copy()
                st = vp.stats()
        else:
            vp(scene, film, cam)
            acc = vp.read_accumulators(film).copy()
            st = vp.stats()
        return acc, (int(st.rays_closest), int(st.rays_shadow), int(st.medium_collisions))
    finally:
        vp.close(trim_cache=True)      # the 72 - 84 GB slab goes back to the driver: later tests size their passes from the free memory


def _check_film(acc, n_pix, spp):
    assert np.isfinite(acc).all() and (acc[:3 * n_pix] >= 0).all() and acc[:3 * n_pix].max() > 0
    w = acc[3 * n_pix:]
    assert np.allclose(w, w[0], rtol=1e-2) and w[0] > 0     # every pixel received the filter weights of its `spp` samples


def test_cornell_bench_pass_equals_split_passes(hk):
    """BASELINE configs[1] exactly as bench.py builds it (SURVEY 8(d)'s two tessellated spheres, 800 x 800, depth 8): 256 spp as ONE pass
    of 164 M paths == 4 x 64 spp, and == the same pass on one stream / without the shared ticket word."""
    from hikari_jl_amd import scenes
    s, film, cam = scenes.cornell_box(800, 800, light="area", objects="two_spheres")
    n_pix = 800 * 800
    one, c_one = _frame(hk, s, film, cam, 8, 256)
    _check_film(one, n_pix, 256)
    assert c_one[0] >= 256 * n_pix and c_one[1] > 0
    split, c_split = _frame(hk, s, film, cam, 8, 256, per_pass=64)
    assert c_one == c_split, (c_one, c_split)
    assert np.array_equal(one, split), float(np.abs(one - split).max())
    for opt in (dict(HK_OVERLAP=0, HK_TICKET_SHARE=0), dict(HK_OVERLAP=1), dict(HK_LEAN_RECORDS=0, HK_SHADOW_FINAL=0)):      # (the last: the record layouts of rounds 3-5)
        again, c_again = _frame(hk, s, film, cam, 8, 256, options=opt)
        assert c_again == c_one, (opt, c_one, c_again)
        assert np.array_equal(one, again), (opt, float(np.abs(one - again).max()))


def test_cloud_bench_frame_equals_split_passes(hk):
    """BASELINE configs[3] as bench.py builds it (1024 x 1024, depth 32, 256 spp = 268 M paths: the library renders 183 + 73 spp) against
    2 x 128 spp: identical ray and collision counters (delta / ratio tracking consume the same random numbers whatever the pass
    size), identical accumulators."""
    from hikari_jl_amd import scenes
    s, film, cam = scenes.bomex_scene(1024, 1024, res=(256, 256, 128), fill=0.05, max_extinction=620.0, majorant_res=(64, 64, 64))
    n_pix = 1024 * 1024
    one, c_one = _frame(hk, s, film, cam, 32, 256)
    _check_film(one, n_pix, 256)
    assert c_one[2] > n_pix
    split, c_split = _frame(hk, s, film, cam, 32, 256, per_pass=128)
    assert c_one == c_split, (c_one, c_split)
    assert np.allclose(one, split, rtol=1e-6, atol=1e-7)
    assert np.array_equal(one, split), float(np.abs(one - split).max())


def test_many_light_bench_pass_equals_split_passes(hk):
    """BASELINE configs[4] stand-in as bench.py builds it (10^6 triangles, 5 x 10^4 lights in the light BVH, 1024 x 1024, depth 8): 256 spp
    in the library's own passes (183 + 73: quantised nodes, pooled light selection, second stream) == 4 x 64 spp."""
    from hikari_jl_amd import scenes
    s, film, cam = scenes.many_light_scene(1024, 1024)
    n_pix = 1024 * 1024
    one, c_one = _frame(hk, s, film, cam, 8, 256)
    _check_film(one, n_pix, 256)
    split, c_split = _frame(hk, s, film, cam, 8, 256, per_pass=64)
    assert c_one == c_split, (c_one, c_split)
    assert np.array_equal(one, split), float(np.abs(one - split).max())

"""denoise! (src/denoise.jl): the oracle against an independent numpy restatement and closed forms (CPU), and the HIP kernels
against the oracle through the C-ABI (GPU).  exp / pow differ by an ulp or two between libm and the device: rtol 2e-5."""
import numpy as np
import pytest

import hikari_jl_amd as hk
import oracle


def _lum(img):
    return (np.float32(0.2126) * img[..., 0] + np.float32(0.7152) * img[..., 1]) + np.float32(0.0722) * img[..., 2]


def _variance_np(fb):
    h, w = fb.shape[:2]
    lum = _lum(fb)
    out = np.zeros((h, w), np.float32)
    for r in range(h):
        for c in range(w):
            vals = [lum[rr, cc] for rr in range(r - 1, r + 2) for cc in range(c - 1, c + 2) if 0 <= rr < h and 0 <= cc < w]
            s = np.float32(0)
            s2 = np.float32(0)
            # the kernel iterates dy (rows) outer, dx (cols) inner
            for v in vals:
                s = np.float32(s + v)
                s2 = np.float32(s2 + np.float32(v * v))
            n = np.float32(len(vals))
            mean, mean_sq = np.float32(s / n), np.float32(s2 / n)
            out[r, c] = max(np.float32(0), np.float32(mean_sq - np.float32(mean * mean)))
    return out


def _atrous_np(fb, normal, depth, var, step, cfg):
    """one pass in float64 (a tolerance reference, not a bit-exact one)"""
    h, w = fb.shape[:2]
    k = np.array([1 / 16, 1 / 4, 3 / 8, 1 / 4, 1 / 16])
    lum = _lum(fb).astype(np.float64)
    out = np.zeros_like(fb, dtype=np.float64)
    for r in range(h):
        for c in range(w):
            sw = 0.0
            acc = np.zeros(3)
            eff = cfg.sigma_color * np.sqrt(var[r, c]) + 1e-4 if (cfg.use_variance and var[r, c] > 0) else cfg.sigma_color
            for dy in range(-2, 3):
                for dx in range(-2, 3):
                    qr, qc = min(max(r + dy * step, 0), h - 1), min(max(c + dx * step, 0), w - 1)
                    wc = np.exp(-abs(lum[r, c] - lum[qr, qc]) / eff)
                    wn = max(0.0, float(np.dot(normal[r, c].astype(np.float64), normal[qr, qc].astype(np.float64)))) ** cfg.sigma_normal
                    with np.errstate(invalid="ignore"):
                        wd = np.exp(-abs(np.float64(depth[r, c]) - np.float64(depth[qr, qc])) / (cfg.sigma_depth * step + 1e-4))
                    wt = k[dx + 2] * k[dy + 2] * wc * wn * wd
                    acc += fb[qr, qc].astype(np.float64) * wt
                    sw += wt
            out[r, c] = acc / sw if sw > 1e-6 else fb[r, c]
    return out


def _scene_buffers(h=18, w=26, seed=3):
    rng = np.random.default_rng(seed)
    fb = (rng.random((h, w, 3)) * 2.0).astype(np.float32)
    normal = np.zeros((h, w, 3), np.float32)
    normal[:, : w // 2] = (0, 0, 1)
    tilt = np.array([0.6, 0.0, 0.8], np.float32)
    normal[:, w // 2:] = tilt
    depth = (2.0 + 0.05 * np.arange(w, dtype=np.float32))[None, :].repeat(h, 0).astype(np.float32)
    depth[:3, :4] = np.inf          # escaped pixels (film.jl:455-460)
    return fb, normal, depth


def test_oracle_variance_and_single_pass_match_numpy():
    oracle.build()
    fb, normal, depth = _scene_buffers()
    cfg = hk.DenoiseConfig(iterations=1, sigma_color=0.7, sigma_normal=8.0, sigma_depth=0.5, use_variance=True)
    out, after = oracle.denoise(cfg.record(), fb, normal, depth)
    var = _variance_np(fb)
    ref = _atrous_np(fb, normal, depth, var, 1, cfg)
    assert np.allclose(out, ref, rtol=2e-5, atol=1e-6)
    assert np.array_equal(after, fb)                       # one pass never writes the framebuffer
    # second pass (step 2) reads pass 1 and lands in the framebuffer
    cfg2 = hk.DenoiseConfig(iterations=2, sigma_color=0.7, sigma_normal=8.0, sigma_depth=0.5, use_variance=True)
    out2, after2 = oracle.denoise(cfg2.record(), fb, normal, depth)
    ref2 = _atrous_np(out, normal, depth, var, 2, cfg2)
    assert np.allclose(out2, ref2, rtol=5e-5, atol=1e-6)
    assert np.array_equal(after2, out2)


def test_oracle_closed_forms():
    oracle.build()
    h, w = 12, 16
    normal = np.zeros((h, w, 3), np.float32)
    normal[..., 1] = 1
    depth = np.full((h, w), 3.0, np.float32)
    const = np.full((h, w, 3), 0.25, np.float32)
    out, _ = oracle.denoise(hk.DenoiseConfig().record(), const, normal, depth)
    assert np.allclose(out, 0.25, rtol=1e-6)               # normalised weights: a constant image is a fixed point
    # orthogonal normals: weight_normal = 0^128 = 0, nothing crosses the edge
    fb = np.zeros((h, w, 3), np.float32)
    fb[:, : w // 2] = 1.0
    normal[:, w // 2:] = (1, 0, 0)
    out, _ = oracle.denoise(hk.DenoiseConfig(use_variance=False).record(), fb, normal, depth)
    assert np.allclose(out[:, : w // 2], 1.0, rtol=1e-6) and np.all(out[:, w // 2:] == 0.0)
    # +Inf centre depth: every weight is NaN, the pixel is kept; iterations = 0 copies the framebuffer
    rng = np.random.default_rng(0)
    fb = rng.random((h, w, 3)).astype(np.float32)
    depth[:] = np.inf
    out, after = oracle.denoise(hk.DenoiseConfig(iterations=3).record(), fb, normal, depth)
    assert np.array_equal(out, fb) and np.array_equal(after, fb)
    out, after = oracle.denoise(hk.DenoiseConfig(iterations=0).record(), fb, normal, np.full((h, w), 1.0, np.float32))
    assert np.array_equal(out, fb)
    # the filter is a convex combination: output stays inside the input range
    out, _ = oracle.denoise(hk.DenoiseConfig().record(), fb, normal, np.full((h, w), 1.0, np.float32))
    assert out.min() >= fb.min() - 1e-6 and out.max() <= fb.max() + 1e-6 and out.std() < fb.std()


@pytest.mark.gpu
@pytest.mark.parametrize("iterations,use_variance", [(0, True), (1, True), (2, False), (5, True)])
def test_gpu_denoise_matches_oracle(iterations, use_variance):
    oracle.build()
    fb, normal, depth = _scene_buffers(40, 56, seed=7)
    cfg = hk.DenoiseConfig(iterations=iterations, sigma_color=1.5, sigma_normal=32.0, sigma_depth=0.7, use_variance=use_variance)
    want, want_after = oracle.denoise(cfg.record(), fb, normal, depth)
    film = hk.Film((56, 40))
    film.framebuffer = fb.copy()
    film.normal, film.depth = normal, depth
    got = film.denoise(cfg)
    assert np.allclose(got, want, rtol=2e-5 * max(iterations, 1), atol=1e-6)
    assert np.allclose(film.framebuffer, want_after, rtol=2e-5 * max(iterations, 1), atol=1e-6)
    if iterations < 2:
        assert np.array_equal(film.framebuffer, fb)


@pytest.mark.gpu
def test_gpu_denoise_rendered_frame():
    """the whole reference flow: render, fill_aux_buffers!, denoise!, postprocess! — noise drops, edges stay"""
    from hikari_jl_amd import scenes
    oracle.build()
    scene, film, cam = scenes.cornell_box(96, 96, light="area")
    vp = hk.VolPath(max_depth=5, samples=8)
    vp(scene, film, cam)
    film.fill_aux_buffers(scene, cam)
    noisy = film.framebuffer.copy()
    want, _ = oracle.denoise(hk.DenoiseConfig().record(), noisy, film.normal, film.depth)
    got = film.denoise()
    assert np.allclose(got, want, rtol=2e-4, atol=1e-6)
    assert np.isfinite(got).all()
    # high-frequency energy (difference to the 4-neighbour mean) drops
    def hf(img):
        m = (img[:-2, 1:-1] + img[2:, 1:-1] + img[1:-1, :-2] + img[1:-1, 2:]) / 4
        return float(np.abs(img[1:-1, 1:-1] - m).mean())
    assert hf(got) < 0.5 * hf(noisy)
    film.denoise_inplace()
    assert np.array_equal(film.framebuffer, film.postprocess_buffer)

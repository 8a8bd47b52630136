"""Second source for the decisions the path takes by HASHING FLOAT BIT PATTERNS — until round 4 pinned only in aggregate (their
fractions, tests/test_independent_pins.py::test_alpha_and_mix_fractions):

  * the MixMaterial choice           mix_hash_float / choose_material            materials/mix-material.jl:116-157, 180-197
  * the stochastic alpha test        pcg32_init(pbrt_hash(o), pbrt_hash(d))      integrators/volpath/intersection.jl:233-252 (camera / bounce rays),
                                     first pcg32_uniform_f32 > alpha                                          :390-396 (shadow rays)
  * their ingredients                MurmurHash64A, pbrt_hash(Vec3f / Point3f), PCG32   materials/spectral-eval.jl:575-636, 700-716, 745-813
  * the ZSobol sampler               Morton index, digit permutations, FastOwen scramble   sampler/sobol.jl:17-323 (every per-pixel comparison rests on it)

restated below in PURE PYTHON INTEGERS (arbitrary precision, masked to 64 bits by hand — nothing is shared with oracle/ or the HIP
library, not even the overflow behaviour of a machine word) from the text of those reference lines, and compared BIT FOR BIT:
against the oracle on the CPU (-m "not gpu") and against the device (-m gpu).  The alpha test is pinned through whole frames: one
sample per pixel through a black cut-out of alpha a in front of an emitter is non-zero exactly where the restated hash lets the
camera ray pass, for the rays each side reports for that pixel (oracle.camera_samples / hk_test_camera)."""
import ctypes as C
import struct

import numpy as np
import pytest

f32 = np.float32
M64 = (1 << 64) - 1


# ---------------------------------------------------------------------------------------------------- the restatement (integers only)
def murmur64a(data: bytes, seed: int = 0) -> int:
    """MurmurHash2, 64-bit version A (Austin Appleby's public-domain MurmurHash64A; spectral-eval.jl:575-636)."""
    m, r = 0xC6A4A7935BD1E995, 47
    n = len(data)
    h = (seed ^ (n * m)) & M64
    full = n // 8
    for i in range(full):
        k = int.from_bytes(data[8 * i:8 * i + 8], "little")
        k = (k * m) & M64
        k ^= k >> r
        k = (k * m) & M64
        h ^= k
        h = (h * m) & M64
    tail = data[8 * full:]
    if tail:                                   # the C switch falls through from the highest remaining byte down to the first, then multiplies
        h ^= int.from_bytes(tail, "little")
        h = (h * m) & M64
    h ^= h >> r
    h = (h * m) & M64
    h ^= h >> r
    return h


def pbrt_hash3(v) -> int:
    """pbrt_hash(::Vec3f) / pbrt_hash(::Point3f): the twelve bytes of the three binary32 values, seed 0 (spectral-eval.jl:700-716)."""
    return murmur64a(struct.pack("<3f", float(v[0]), float(v[1]), float(v[2])), 0)


PCG_MULT = 0x5851F42D4C957F2D


def pcg32_first_float(seq: int, seed: int) -> np.float32:
    """pcg32_init(seq, seed) followed by ONE pcg32_uniform_f32 (spectral-eval.jl:767-813)."""
    inc = ((seq << 1) | 1) & M64
    state = (0 * PCG_MULT + inc) & M64
    state = (state + seed) & M64
    state = (state * PCG_MULT + inc) & M64
    old = state
    xs = (((old >> 18) ^ old) >> 27) & 0xFFFFFFFF
    rot = (old >> 59) & 31
    u = ((xs >> rot) | (xs << ((32 - rot) & 31))) & 0xFFFFFFFF
    f = f32(u) * f32(2.3283064e-10)            # Float32(u32) * 2.3283064f-10, rounded to binary32 twice (conversion, product)
    lim = f32(1.0) - np.finfo(f32).eps
    return f if f < lim else lim


def alpha_u(o, d) -> np.float32:
    return pcg32_first_float(pbrt_hash3(o), pbrt_hash3(d))


def bits(x) -> int:
    return struct.unpack("<I", struct.pack("<f", float(x)))[0]


def mix_hash_float(p, wo, key) -> np.float32:
    """mix-material.jl:116-157; key = (type index, vector index) of the two children's SetKeys.  A UInt32 shifted left stays a UInt32
    (Julia: the high bits fall off) before it is widened by the xor with the UInt64 accumulator."""
    c1, c2 = 0xCC9E2D51, 0x1B873593
    h = 0
    h ^= bits(p[0])
    h = (h * c1) & M64
    h ^= (bits(p[1]) << 4) & 0xFFFFFFFF
    h = (h * c2) & M64
    h ^= (bits(p[2]) << 8) & 0xFFFFFFFF
    h ^= (bits(wo[0]) << 16) & 0xFFFFFFFF
    h = (h * c1) & M64
    h ^= bits(wo[1])
    h = (h * c2) & M64
    h ^= (bits(wo[2]) << 12) & 0xFFFFFFFF
    h ^= (int(key[0]) << 24) & M64           # UInt64(type_idx) << 24: widened first
    h ^= int(key[1])
    h = (h * c1) & M64
    h ^= (int(key[2]) << 28) & M64
    h ^= (int(key[3]) << 4) & M64
    h = (h * c2) & M64
    h ^= h >> 31
    h = (h * 0x7FB5D329728EA185) & M64
    h ^= h >> 27
    h = (h * 0x81DADEF4BC2DD44D) & M64
    h ^= h >> 33
    return f32(h & 0xFFFFFFFF) * f32(2.0 ** -32)


# ZSobol (sampler/sobol.jl:17-323): the sampler every per-pixel comparison rests on — integers only as well
M32 = (1 << 32) - 1
PERMUTATIONS_4WAY = ((0, 1, 2, 3), (0, 1, 3, 2), (0, 2, 1, 3), (0, 2, 3, 1), (0, 3, 2, 1), (0, 3, 1, 2), (1, 0, 2, 3), (1, 0, 3, 2), (1, 2, 0, 3), (1, 2, 3, 0), (1, 3, 2, 0),
                     (1, 3, 0, 2), (2, 1, 0, 3), (2, 1, 3, 0), (2, 0, 1, 3), (2, 0, 3, 1), (2, 3, 0, 1), (2, 3, 1, 0), (3, 1, 2, 0), (3, 1, 0, 2), (3, 2, 1, 0), (3, 2, 0, 1),
                     (3, 0, 2, 1), (3, 0, 1, 2))          # sobol.jl:159-184 (pbrt-v4 samplers.h)


def mix_bits64(v: int) -> int:                            # spectral-eval.jl:641-648
    v ^= v >> 31
    v = (v * 0x7FB5D329728EA185) & M64
    v ^= v >> 27
    v = (v * 0x81DADEF4BC2DD44D) & M64
    v ^= v >> 33
    return v


def spread_bits(x: int) -> int:                           # left_shift2, sobol.jl:42-50
    x &= 0xFFFFFFFF
    x = (x ^ (x << 16)) & 0x0000FFFF0000FFFF
    x = (x ^ (x << 8)) & 0x00FF00FF00FF00FF
    x = (x ^ (x << 4)) & 0x0F0F0F0F0F0F0F0F
    x = (x ^ (x << 2)) & 0x3333333333333333
    x = (x ^ (x << 1)) & 0x5555555555555555
    return x


def bitreverse32(v: int) -> int:
    return int("{:032b}".format(v)[::-1], 2)


def fast_owen_scramble(v: int, seed: int) -> int:         # sobol.jl:73-81
    v = bitreverse32(v)
    v ^= (v * 0x3D20ADEA) & M32
    v = (v + seed) & M32
    v = (v * ((seed >> 16) | 1)) & M32
    v ^= (v * 0x05526C56) & M32
    v ^= (v * 0x53A22864) & M32
    return bitreverse32(v)


def zsobol_params(spp: int, width: int, height: int):     # compute_zsobol_params, sobol.jl:316-322
    log2_spp = (max(1, spp) - 1).bit_length()
    res_log2 = (max(width, height) - 1).bit_length()
    return log2_spp, res_log2 + (log2_spp + 1) // 2


def zsobol_sample_index(morton: int, dim: int, log2_spp: int, n_digits: int) -> int:     # sobol.jl:211-262
    odd = log2_spp & 1
    index = 0
    for i in range(n_digits - 1, odd - 1, -1):            # (the reference runs 32 masked iterations; those below `last_digit` contribute nothing)
        shift = max(0, 2 * i - odd)
        digit = (morton >> shift) & 3
        higher = morton >> (shift + 2)
        p = (mix_bits64(higher ^ ((0x55555555 * dim) & M64)) >> 24) % 24
        index |= PERMUTATIONS_4WAY[p][digit] << shift
    if odd:
        index |= ((morton & 1) ^ (mix_bits64((morton >> 1) ^ ((0x55555555 * dim) & M64)) & 1))
    return index


def sobol_sample(a: int, dimension: int, seed: int, matrices) -> np.float32:             # sobol.jl:100-126
    v = 0
    for bit in range(52):
        if (a >> bit) & 1:
            v ^= int(matrices[dimension * 52 + bit])
    v = fast_owen_scramble(v, seed)
    f = f32(v) * f32(2.3283064365386963e-10)
    lim = f32(1.0) - np.finfo(f32).eps
    return f if f < lim else lim


def zsobol(px, py, sample_idx, dim, log2_spp, n_digits, seed, matrices, two=False):      # sobol.jl:274-311
    morton = ((((spread_bits(py) << 1) | spread_bits(px)) << log2_spp) | sample_idx) & M64
    index = zsobol_sample_index(morton, dim, log2_spp, n_digits)
    h = murmur64a(struct.pack("<iI", dim + (2 if two else 1), seed), 0)      # Hash(dimension AFTER the increment, seed)
    if two:
        return sobol_sample(index, 0, h & M32, matrices), sobol_sample(index, 1, h >> 32, matrices)
    return sobol_sample(index, 0, h & M32, matrices)


def _unit(v):
    return (v / np.linalg.norm(v, axis=1, keepdims=True)).astype(f32)


def _pf(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _pi(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


# ---------------------------------------------------------------------------------------------------- CPU: the oracle against the restatement
def test_murmur_and_pcg_restated_in_integers(oracle):
    """10^5 random 12-byte keys (and a few other lengths, tails of 1..7 bytes included): MurmurHash64A of the oracle == the integer
    restatement; 2 * 10^4 (sequence, seed) pairs: the first float of the PCG32 stream, bit for bit."""
    L = oracle.lib()
    L.hko_murmur64a.restype = C.c_uint64
    L.hko_murmur64a.argtypes = [C.c_char_p, C.c_int32, C.c_uint64]
    assert murmur64a(b"", 0) == 0                                  # h = 0 ^ 0, every step keeps 0
    rng = np.random.default_rng(401)
    raw = rng.integers(0, 256, size=(100000, 12), dtype=np.uint8)
    for i in range(raw.shape[0]):
        b = raw[i].tobytes()
        assert L.hko_murmur64a(b, 12, 0) == murmur64a(b, 0), i
    for n in (1, 3, 4, 7, 8, 9, 15, 16, 20, 23):
        for _ in range(200):
            b = rng.integers(0, 256, size=n, dtype=np.uint8).tobytes()
            seed = int(rng.integers(0, 1 << 63))
            assert L.hko_murmur64a(b, n, seed) == murmur64a(b, seed), (n, seed)
    u32 = np.empty(1, np.uint32)
    fl = np.empty(1, f32)
    for _ in range(20000):
        seq, seed = int(rng.integers(0, 1 << 63)) * 2 + int(rng.integers(0, 2)), int(rng.integers(0, 1 << 63)) * 2 + int(rng.integers(0, 2))
        L.hko_pcg32(C.c_uint64(seq), C.c_uint64(seed), 1, 1, u32.ctypes.data_as(C.POINTER(C.c_uint32)), _pf(fl))
        assert fl[0].view(np.uint32) == np.asarray(pcg32_first_float(seq, seed), f32).view(np.uint32), (seq, seed)


@pytest.mark.parametrize("width,height,spp", [(800, 800, 256), (64, 48, 16), (1024, 1024, 8), (33, 70, 4096), (16, 16, 1)])
def test_zsobol_restated_in_integers(hk, oracle, width, height, spp):
    """ZSobolSampler (Morton index, base-4 digit permutations hashed from the higher digits, FastOwen scrambling; power-of-4 and
    odd-power-of-2 sample counts): 4 000 random (pixel, sample index, dimension) draws per film / sample-count combination, 1-D and 2-D,
    the oracle's float bit for bit."""
    mats = hk.tables.load()["sobol"]
    rng = np.random.default_rng(403 + width)
    n = 4000
    px = rng.integers(1, width + 1, n).astype(np.int32)
    py = rng.integers(1, height + 1, n).astype(np.int32)
    eff = max(spp, 4096)                                      # render! sizes the index for at least 4 096 samples (volpath.jl:474-479)
    si = rng.integers(1, min(eff, 1 << 14) + 1, n).astype(np.int32)
    dim = rng.integers(0, 110, n).astype(np.int32)
    seed = 0
    o1, o2 = oracle.sobol(width, height, eff, seed, px, py, si, dim)
    log2_spp, n_digits = zsobol_params(eff, width, height)
    for i in range(n):
        a = zsobol(int(px[i]), int(py[i]), int(si[i]), int(dim[i]), log2_spp, n_digits, seed, mats)
        b0, b1 = zsobol(int(px[i]), int(py[i]), int(si[i]), int(dim[i]), log2_spp, n_digits, seed, mats, two=True)
        assert np.asarray(a, f32).view(np.uint32) == o1[i].view(np.uint32), (i, a, o1[i])
        assert np.asarray(b0, f32).view(np.uint32) == o2[i, 0].view(np.uint32) and np.asarray(b1, f32).view(np.uint32) == o2[i, 1].view(np.uint32), (i, b0, b1, o2[i])


def _mix_scene(hk, amount):
    from hikari_jl_amd import geometry as G
    R = hk.RGBSpectrum
    s = hk.Scene()
    mix = hk.MixMaterial((hk.MatteMaterial(Kd=R(0.8, 0.1, 0.1)), hk.MirrorMaterial(Kr=R(0.9))), amount=amount)
    s.push(G.quad((-1, -1, 0), (1, -1, 0), (1, 1, 0), (-1, 1, 0), normal=(0, 0, 1)), mix)
    s.sync()
    kinds = [s.desc.materials[i].kind for i in range(s.desc.n_materials)]
    mi = kinds.index(hk._abi.HK_MAT_MIX)
    rec = s.desc.materials[mi]
    return s, mi, tuple(int(k) for k in rec.mix_key), (int(rec.i[0]), int(rec.i[1]))


def _mix_inputs(n):
    rng = np.random.default_rng(402)
    p = (rng.random((n, 3)) * 2 - 1).astype(f32)
    wo = _unit(rng.normal(size=(n, 3)))
    p[:64] = 0.0                                                     # +0 / -0 / denormal / huge bit patterns go through the same shifts
    p[64:128] = -0.0
    p[128:192] = f32(1e-41)
    wo[192:256] = f32(3e38)
    uv = rng.random((n, 2), dtype=f32)
    return p, wo, uv


def _restated_choice(p, wo, key, children, amount):
    """choose_material (mix-material.jl:180-197) for a constant amount in (0, 1): amount < u ? first : second."""
    a = f32(amount)
    return np.array([children[0] if a < mix_hash_float(p[i], wo[i], key) else children[1] for i in range(p.shape[0])], np.int32)


@pytest.mark.parametrize("amount", [0.2, 0.5, 0.85])
def test_mix_choice_restated_in_integers(hk, oracle, amount):
    """10^5 points per amount: the material index the oracle resolves == the integer restatement's choice, for every point."""
    s, mi, key, children = _mix_scene(hk, amount)
    p, wo, uv = _mix_inputs(100000)
    want = _restated_choice(p, wo, key, children, amount)
    osc = oracle.OracleScene(s)
    got = osc.mix_resolve(mi, p, wo, uv)
    osc.close()
    assert np.array_equal(got, want), int((got != want).sum())
    assert 0.9 * amount < (want == children[1]).mean() < 1.1 * amount + 0.01      # (and it is a fair coin of weight `amount`)


def _cutout_scene(hk, alpha, wh):
    """camera at the origin looking along +z at a big emitter (z = 6) through a BLACK matte cut-out of constant alpha (z = 3)"""
    from hikari_jl_amd import geometry as G
    from hikari_jl_amd.materials import Texture
    R = hk.RGBSpectrum
    s = hk.Scene()
    em = G.quad((-6, -6, 6), (-6, 6, 6), (6, 6, 6), (6, -6, 6), normal=(0, 0, -1))
    s.push(em, hk.MediumInterface(hk.MatteMaterial(Kd=R(0.0)), emission=hk.Emissive(Le=R(0.2), scale=1.0, two_sided=True)))
    tex = np.zeros((4, 4, 4), f32)
    tex[..., 3] = alpha
    s.push(G.quad((-2, -2, 3), (2, -2, 3), (2, 2, 3), (-2, 2, 3), normal=(0, 0, -1)), hk.MatteMaterial(Kd=Texture(tex)))
    s.sync()
    film = hk.Film((wh, wh))
    cam = hk.PerspectiveCamera((0, 0, 0), (0, 0, 1), film, fov=20.0)
    return s, film, cam


def _pixels(wh):
    py, px = np.meshgrid(np.arange(1, wh + 1), np.arange(1, wh + 1), indexing="ij")
    return px.ravel().astype(np.int32), py.ravel().astype(np.int32)


def _lit(img):
    return (np.asarray(img, np.float64).sum(axis=2) > 0.0)


@pytest.mark.parametrize("alpha", [0.3, 0.7])
def test_alpha_decision_per_pixel_oracle(hk, oracle, alpha):
    """One sample per pixel: the oracle's frame is lit exactly in the pixels whose camera ray (as the oracle's K1 reports it) the
    restated hash lets through the cut-out (intersection.jl:233-252)."""
    wh, sample = 48, 1
    s, film, cam = _cutout_scene(hk, alpha, wh)
    p = hk.integrator_params(max_depth=3, samples=64, max_component_value=1e9)
    px, py = _pixels(wh)
    rays = oracle.camera_samples(p, cam, wh, wh, px, py, np.full(px.shape, sample, np.int32))
    passes = np.array([alpha_u(r[9:12], r[12:15]) > f32(alpha) for r in rays]).reshape(wh, wh)      # [py - 1, px - 1]
    osc = oracle.OracleScene(s)
    acc, _ = osc.render(p, cam, wh, wh, 1, first=sample)
    osc.close()
    lit = _lit(oracle.finalize(acc, wh, wh))
    assert lit.shape == passes.shape
    assert np.array_equal(lit, passes), int((lit != passes).sum())
    assert abs(passes.mean() - (1.0 - alpha)) < 0.05


# ---------------------------------------------------------------------------------------------------- GPU: the device against the restatement
@pytest.mark.gpu
@pytest.mark.parametrize("amount", [0.2, 0.5, 0.85])
def test_mix_choice_device_against_integers(hk, gpu_ctx, amount):
    """hk_test_mix (resolve_mix_material on the device) == the integer restatement, 10^5 points per amount."""
    s, mi, key, children = _mix_scene(hk, amount)
    p, wo, uv = _mix_inputs(100000)
    want = _restated_choice(p, wo, key, children, amount)
    sh = hk.scene_handle(gpu_ctx, s)
    out = np.empty(p.shape[0], np.int32)
    hk._lib.check(hk._lib.lib().hk_test_mix(gpu_ctx.h, sh, mi, p.shape[0], _pf(p), _pf(wo), _pf(uv), _pi(out)), "hk_test_mix")
    assert np.array_equal(out, want), int((out != want).sum())


@pytest.mark.gpu
@pytest.mark.parametrize("alpha", [0.3, 0.7])
def test_alpha_decision_per_pixel_device(hk, gpu_ctx, alpha):
    """The device's one-sample frame is lit exactly where the restated hash lets the DEVICE's camera ray (hk_test_camera) pass: the alpha
    loop of k_trace seeds its PCG from the bits of the ray it casts."""
    wh, sample = 48, 1
    s, film, cam = _cutout_scene(hk, alpha, wh)
    p = hk.integrator_params(max_depth=3, samples=64, max_component_value=1e9)
    px, py = _pixels(wh)
    L = hk._lib.lib()
    integ = C.c_void_p()
    hk._lib.check(L.hk_integrator_create(gpu_ctx.h, C.byref(p), C.byref(integ)), "hk_integrator_create")
    rays = np.empty((px.shape[0], 15), f32)
    rec = cam.record()
    hk._lib.check(L.hk_test_camera(gpu_ctx.h, integ, C.byref(rec), wh, wh, px.shape[0], _pi(px), _pi(py), _pi(np.full(px.shape, sample, np.int32)), _pf(rays)), "hk_test_camera")
    L.hk_integrator_destroy(integ)
    passes = np.array([alpha_u(r[9:12], r[12:15]) > f32(alpha) for r in rays]).reshape(wh, wh)
    vp = hk.VolPath(max_depth=3, samples=64, max_component_value=1e9)
    vp._ensure(film)
    vp.clear()
    vp.render_samples(s, film, cam, 1, first=sample)
    lit = _lit(film.framebuffer)
    vp.close()
    assert np.array_equal(lit, passes), int((lit != passes).sum())
    assert abs(passes.mean() - (1.0 - alpha)) < 0.05


@pytest.mark.gpu
def test_alpha_decision_of_shadow_rays_device(hk, oracle, gpu_ctx):
    """Shadow rays take the same test with THEIR origin and direction (intersection.jl:390-396): a matte floor lit by a point light through
    the cut-out.  The shadow ray's origin is the shading point, known only to within the intersection arithmetic, so this is pinned against
    the oracle's frame (same rays bit for bit in this scene: closest hits are bit-exact) instead of the restatement: identical lit / unlit
    pattern of the direct light, one sample per pixel, pixel by pixel."""
    from hikari_jl_amd import geometry as G
    from hikari_jl_amd.materials import Texture
    R = hk.RGBSpectrum
    wh = 40
    s = hk.Scene()
    s.push(G.quad((-2, 0, -2), (2, 0, -2), (2, 0, 2), (-2, 0, 2), normal=(0, 1, 0)), hk.MatteMaterial(Kd=R(0.7)))
    tex = np.zeros((4, 4, 4), f32)
    tex[..., 3] = 0.5
    s.push(G.quad((-3, 1, -3), (3, 1, -3), (3, 1, 3), (-3, 1, 3), normal=(0, -1, 0)), hk.MatteMaterial(Kd=Texture(tex)))
    s.push(hk.PointLight.from_spectrum_first(R(8.0), (0.0, 2.0, 0.0)))
    s.sync()
    film = hk.Film((wh, wh))
    cam = hk.PerspectiveCamera((0, 0.8, -1.9), (0, 0.0, 0.2), film, fov=50.0)
    p = hk.integrator_params(max_depth=1, samples=64, max_component_value=1e9)
    osc = oracle.OracleScene(s)
    acc, _ = osc.render(p, cam, wh, wh, 1, first=1)
    osc.close()
    want = _lit(oracle.finalize(acc, wh, wh))
    vp = hk.VolPath(max_depth=1, samples=64, max_component_value=1e9)
    vp._ensure(film)
    vp.clear()
    vp.render_samples(s, film, cam, 1, first=1)
    got = _lit(film.framebuffer)
    vp.close()
    assert 0.2 < want.mean() < 0.8
    assert (got != want).mean() < 0.002, float((got != want).mean())      # (a pixel may differ where the camera ray's direction differs by an ulp)

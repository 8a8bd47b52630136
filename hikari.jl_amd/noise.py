"""Procedural noise of Hikari's `src/random.jl` on whole arrays: `perlin3d` (:36-49), `fbm3d` (:64-74), `worley3d` (:86-112),
`worley_fbm3d` (:119-129) and `generate_cloud_density` (:154-218).  Host-side scene building only — the BOMEX stand-in of
BASELINE.json configs[3] is made with it (`scenes.cloud_density`).  Same arithmetic as the reference (float64 noise on float32
voxel-centre coordinates), evaluated slab by slab instead of voxel by voxel."""
import numpy as np

# Ken Perlin's reference permutation (random.jl:16-33): a published constant table
_PERM = np.array([
    151, 160, 137, 91, 90, 15, 131, 13, 201, 95, 96, 53, 194, 233, 7, 225, 140, 36, 103, 30, 69, 142,
    8, 99, 37, 240, 21, 10, 23, 190, 6, 148, 247, 120, 234, 75, 0, 26, 197, 62, 94, 252, 219, 203, 117,
    35, 11, 32, 57, 177, 33, 88, 237, 149, 56, 87, 174, 20, 125, 136, 171, 168, 68, 175, 74, 165, 71,
    134, 139, 48, 27, 166, 77, 146, 158, 231, 83, 111, 229, 122, 60, 211, 133, 230, 220, 105, 92, 41,
    55, 46, 245, 40, 244, 102, 143, 54, 65, 25, 63, 161, 1, 216, 80, 73, 209, 76, 132, 187, 208, 89,
    18, 169, 200, 196, 135, 130, 116, 188, 159, 86, 164, 100, 109, 198, 173, 186, 3, 64, 52, 217, 226,
    250, 124, 123, 5, 202, 38, 147, 118, 126, 255, 82, 85, 212, 207, 206, 59, 227, 47, 16, 58, 17, 182,
    189, 28, 42, 223, 183, 170, 213, 119, 248, 152, 2, 44, 154, 163, 70, 221, 153, 101, 155, 167, 43,
    172, 9, 129, 22, 39, 253, 19, 98, 108, 110, 79, 113, 224, 232, 178, 185, 112, 104, 218, 246, 97,
    228, 251, 34, 242, 193, 238, 210, 144, 12, 191, 179, 162, 241, 81, 51, 145, 235, 249, 14, 239,
    107, 49, 192, 214, 31, 181, 199, 106, 157, 184, 84, 204, 176, 115, 121, 50, 45, 127, 4, 150, 254,
    138, 236, 205, 93, 222, 114, 67, 29, 24, 72, 243, 141, 128, 195, 78, 66, 215, 61, 156, 180], dtype=np.int64)


def _perm(i):
    return _PERM[i & 255]


def _fade(t):
    return t * t * t * (t * (t * 6 - 15) + 10)


def _lerp(t, a, b):
    return a + t * (b - a)


def _grad(h, x, y, z):
    h = h & 15
    u = np.where(h < 8, x, y)
    v = np.where(h < 4, y, np.where((h == 12) | (h == 14), x, z))
    return np.where((h & 1) == 0, u, -u) + np.where((h & 2) == 0, v, -v)


def perlin3d(x, y, z):
    """Classic Perlin noise on broadcastable float64 arrays (random.jl:36-49)."""
    x, y, z = np.broadcast_arrays(np.asarray(x, np.float64), np.asarray(y, np.float64), np.asarray(z, np.float64))
    fx, fy, fz = np.floor(x), np.floor(y), np.floor(z)
    X, Y, Z = fx.astype(np.int64) & 255, fy.astype(np.int64) & 255, fz.astype(np.int64) & 255
    x, y, z = x - fx, y - fy, z - fz
    u, v, w = _fade(x), _fade(y), _fade(z)
    A, B = _perm(X) + Y, _perm(X + 1) + Y
    AA, AB, BA, BB = _perm(A) + Z, _perm(A + 1) + Z, _perm(B) + Z, _perm(B + 1) + Z
    return _lerp(w,
                 _lerp(v, _lerp(u, _grad(_perm(AA), x, y, z), _grad(_perm(BA), x - 1, y, z)),
                       _lerp(u, _grad(_perm(AB), x, y - 1, z), _grad(_perm(BB), x - 1, y - 1, z))),
                 _lerp(v, _lerp(u, _grad(_perm(AA + 1), x, y, z - 1), _grad(_perm(BA + 1), x - 1, y, z - 1)),
                       _lerp(u, _grad(_perm(AB + 1), x, y - 1, z - 1), _grad(_perm(BB + 1), x - 1, y - 1, z - 1))))


def fbm3d(x, y, z, octaves=4, persistence=0.5):
    """random.jl:64-74"""
    total, frequency, amplitude, max_value = 0.0, 1.0, 1.0, 0.0
    for _ in range(octaves):
        total = total + perlin3d(x * frequency, y * frequency, z * frequency) * amplitude
        max_value += amplitude
        amplitude *= persistence
        frequency *= 2.0
    return total / max_value


def worley3d(x, y, z, seed=0):
    """Distance to the nearest feature point of the 27 neighbouring cells (random.jl:86-112)."""
    x, y, z = np.broadcast_arrays(np.asarray(x, np.float64), np.asarray(y, np.float64), np.asarray(z, np.float64))
    xi, yi, zi = np.floor(x).astype(np.int64), np.floor(y).astype(np.int64), np.floor(z).astype(np.int64)
    fx, fy, fz = x - xi, y - yi, z - zi
    best = np.full(x.shape, 100.0)                      # squared distances; the reference's 10.0 start never survives
    for dz in (-1, 0, 1):
        hz = (zi + dz) & 255
        for dy in (-1, 0, 1):
            hy = (yi + dy) & 255
            for dx in (-1, 0, 1):
                h = _perm(_perm(_perm((xi + dx + seed) & 255) + hy) + hz)
                ddx = fx - (dx + (h & 63) / 64.0)
                ddy = fy - (dy + ((h >> 2) & 63) / 64.0)
                ddz = fz - (dz + ((h >> 4) & 63) / 64.0)
                np.minimum(best, ddx * ddx + ddy * ddy + ddz * ddz, out=best)
    return np.sqrt(best)


def worley_fbm3d(x, y, z, octaves=3, persistence=0.5, lacunarity=2.0):
    """random.jl:119-129 (octave i uses seed 17 i)"""
    total, frequency, amplitude, max_value = 0.0, 1.0, 1.0, 0.0
    for i in range(1, octaves + 1):
        total = total + worley3d(x * frequency, y * frequency, z * frequency, seed=i * 17) * amplitude
        max_value += amplitude
        amplitude *= persistence
        frequency *= lacunarity
    return total / max_value


def cloud_noise_base(resolution, scale=4.0, worley_weight=0.6, slab=4):
    """The `base` field of generate_cloud_density (random.jl:176-190: inverted Worley cells + billowed Perlin + fine turbulence) at
    the voxel centres of an [nx, ny, nz] grid; each axis spans [0, 1] like the reference's cubic grid."""
    nx, ny, nz = (resolution,) * 3 if np.isscalar(resolution) else resolution
    f32 = np.float32
    xs = ((np.arange(1, nx + 1, dtype=f32) - f32(0.5)) / f32(nx)).astype(np.float64)[:, None, None]
    ys = ((np.arange(1, ny + 1, dtype=f32) - f32(0.5)) / f32(ny)).astype(np.float64)[None, :, None]
    zs_all = ((np.arange(1, nz + 1, dtype=f32) - f32(0.5)) / f32(nz)).astype(np.float64)
    base = np.empty((nx, ny, nz), np.float64)

    def one(k):
        zs = zs_all[None, None, k:k + slab]
        worley = 1.0 - worley_fbm3d(xs * scale * 0.8, ys * scale * 0.8, zs * scale * 0.8, octaves=3)
        billow = 1.0 - np.abs(fbm3d(xs * scale * 1.5, ys * scale * 1.5, zs * scale * 1.5, octaves=4, persistence=0.55))
        b = worley_weight * worley + (1.0 - worley_weight) * billow
        b = b + fbm3d(xs * scale * 4.0 + 13.7, ys * scale * 4.0 - 5.3, zs * scale * 4.0 + 9.1, octaves=3) * 0.12
        base[:, :, k:k + slab] = b

    # slabs are independent and numpy releases the GIL inside its loops: one thread per host core the process may use
    import os
    from concurrent.futures import ThreadPoolExecutor
    try:
        n_thr = len(os.sched_getaffinity(0))
    except AttributeError:
        n_thr = os.cpu_count() or 1
    with ThreadPoolExecutor(max(1, min(n_thr, 32))) as ex:
        list(ex.map(one, range(0, nz, slab)))
    return base, (xs, ys, zs_all)


def generate_cloud_density(resolution, scale=4.0, sphere_falloff=True, threshold=0.3, worley_weight=0.6, edge_sharpness=1.5,
                           density_scale=3.0):
    """generate_cloud_density(resolution; ...) -> Float32[nx, ny, nz]  (random.jl:154-218).  `resolution` may also be a triple."""
    base, (xs, ys, zs) = cloud_noise_base(resolution, scale, worley_weight)
    val = np.clip((base - threshold) / (1.0 - threshold), 0.0, 1.0)
    if not sphere_falloff:
        return (val * density_scale).astype(np.float32)
    zs = zs[None, None, :]
    c = np.float64(np.float32(0.5))
    dist = np.sqrt((xs - c) ** 2 + (ys - c) ** 2 + (zs - c) ** 2)
    bn = 0.15 * fbm3d(xs * scale * 2.0 + 7.1, ys * scale * 2.0, zs * scale * 2.0 - 3.3, octaves=3)
    radius = np.float64(np.float32(0.45)) * (1.0 + bn)
    t = dist / radius
    with np.errstate(invalid="ignore", divide="ignore"):
        edge = np.clip(1.0 - (t / (0.3 + 0.7 * base)) ** edge_sharpness, 0.0, 1.0)
    out = np.where(dist < radius, val * edge * density_scale, 0.0)
    return np.nan_to_num(out).astype(np.float32)

"""Host-side mirror of Hikari's participating media records (src/integrators/volpath/media.jl:762-776, 873-935;
nanovdb.jl:153-191) and their host-side builders: `build_majorant_grid` (media.jl:1459-1487),
`build_nanovdb_from_dense` (nanovdb.jl:602-858) and `build_nanovdb_majorant_grid` (nanovdb.jl:1174-1235).
These run once per scene on the host (the reference builds them on the CPU too); the per-ray work
(delta tracking, ratio tracking, NanoVDB tree walks) is in the HIP library."""
import ctypes as C

import numpy as np

from . import _abi as A
from .materials import RGBSpectrum

f32 = np.float32


class Medium:
    kind = -1

    def fill_record(self, rec, keep):
        raise NotImplementedError


def _identity16():
    return [1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1]


def _base_record(rec, kind, sigma_a, sigma_s, Le, g):
    rec.kind = kind
    rec.sigma_a[:] = sigma_a.c
    rec.sigma_s[:] = sigma_s.c
    rec.Le[:] = Le.c
    rec.g = float(f32(g))
    rec.sigma_scale = rec.Le_scale = 1.0
    rec.render_to_medium[:] = _identity16()
    rec.medium_to_render[:] = _identity16()


class HomogeneousMedium(Medium):
    """HomogeneousMedium(; σ_a=0.01, σ_s=1, Le=0, g=0)  (media.jl:762-776)"""
    kind = A.HK_MEDIUM_HOMOGENEOUS

    def __init__(self, sigma_a=RGBSpectrum(0.01), sigma_s=RGBSpectrum(1.0), Le=RGBSpectrum(0.0), g=0.0):
        self.sigma_a, self.sigma_s, self.Le, self.g = sigma_a, sigma_s, Le, g

    def fill_record(self, rec, keep):
        _base_record(rec, self.kind, self.sigma_a, self.sigma_s, self.Le, self.g)


def build_majorant_grid(density, res):
    """media.jl:1459-1487.  density[nx, ny, nz]; returns float32[rx*ry*rz] indexed x + rx*(y + ry*z)."""
    nx, ny, nz = density.shape
    rx, ry, rz = res
    out = np.zeros(rx * ry * rz, dtype=f32)
    for iz in range(rz):
        z0 = max(1, int(np.floor(iz * nz / rz)) + 1)
        z1 = min(nz, int(np.ceil((iz + 1) * nz / rz)))
        for iy in range(ry):
            y0 = max(1, int(np.floor(iy * ny / ry)) + 1)
            y1 = min(ny, int(np.ceil((iy + 1) * ny / ry)))
            for ix in range(rx):
                x0 = max(1, int(np.floor(ix * nx / rx)) + 1)
                x1 = min(nx, int(np.ceil((ix + 1) * nx / rx)))
                blk = density[x0 - 1:x1, y0 - 1:y1, z0 - 1:z1]
                out[ix + rx * (iy + ry * iz)] = max(0.0, float(blk.max())) if blk.size else 0.0
    return out


class GridMedium(Medium):
    """GridMedium(density; σ_a, σ_s, g, bounds, transform, majorant_res=(16,16,16))  (media.jl:873-935)."""
    kind = A.HK_MEDIUM_GRID

    def __init__(self, density, sigma_a=RGBSpectrum(0.01), sigma_s=RGBSpectrum(1.0), g=0.0, bounds=((0, 0, 0), (1, 1, 1)),
                 transform=None, majorant_res=(16, 16, 16)):
        self.density = np.ascontiguousarray(density, dtype=f32)        # [nx, ny, nz]
        self.sigma_a, self.sigma_s, self.g = sigma_a, sigma_s, g
        self.bounds = (tuple(float(f32(v)) for v in bounds[0]), tuple(float(f32(v)) for v in bounds[1]))
        self.medium_to_render = np.eye(4, dtype=f32) if transform is None else np.asarray(transform, dtype=f32)
        self.render_to_medium = np.linalg.inv(self.medium_to_render.astype(np.float64)).astype(f32)
        self.majorant_res = tuple(int(v) for v in majorant_res)
        self.majorant = build_majorant_grid(self.density, self.majorant_res)
        self.max_density = float(self.density.max())

    def fill_record(self, rec, keep):
        _base_record(rec, self.kind, self.sigma_a, self.sigma_s, RGBSpectrum(0.0), self.g)
        rec.bounds_min[:] = self.bounds[0]
        rec.bounds_max[:] = self.bounds[1]
        rec.render_to_medium[:] = [float(x) for x in self.render_to_medium.reshape(-1)]
        rec.medium_to_render[:] = [float(x) for x in self.medium_to_render.reshape(-1)]
        rec.res[:] = self.density.shape
        jl = np.ascontiguousarray(np.transpose(self.density, (2, 1, 0)))   # C [z][y][x] == Julia [x,y,z] column-major
        keep.append(jl)
        rec.density = jl.ctypes.data_as(A.PF)
        rec.majorant_res[:] = self.majorant_res
        rec.majorant = self.majorant.ctypes.data_as(A.PF)
        rec.max_density = self.max_density


def build_rgb_majorant_grid(sigma_a_grid, sigma_s_grid, sigma_scale, grid_size, res):
    """build_rgb_majorant_grid (media.jl:1122-1183): sigma_scale * (max sigma_a component + max sigma_s component) per
    coarse voxel; an absent grid counts as 1.  Returns float32[rx*ry*rz] indexed x + rx*(y + ry*z)."""
    nx, ny, nz = grid_size
    rx, ry, rz = res
    out = np.zeros(rx * ry * rz, dtype=f32)
    amax = sigma_a_grid[..., :3].max(axis=3) if sigma_a_grid is not None else None
    smax = sigma_s_grid[..., :3].max(axis=3) if sigma_s_grid is not None else None
    for iz in range(rz):
        z0 = max(1, int(np.floor(iz * nz / rz)) + 1)
        z1 = min(nz, int(np.ceil((iz + 1) * nz / rz)))
        for iy in range(ry):
            y0 = max(1, int(np.floor(iy * ny / ry)) + 1)
            y1 = min(ny, int(np.ceil((iy + 1) * ny / ry)))
            for ix in range(rx):
                x0 = max(1, int(np.floor(ix * nx / rx)) + 1)
                x1 = min(nx, int(np.ceil((ix + 1) * nx / rx)))
                ma = f32(1.0) if amax is None else f32(max(0.0, float(amax[x0 - 1:x1, y0 - 1:y1, z0 - 1:z1].max(initial=0.0))))
                ms = f32(1.0) if smax is None else f32(max(0.0, float(smax[x0 - 1:x1, y0 - 1:y1, z0 - 1:z1].max(initial=0.0))))
                out[ix + rx * (iy + ry * iz)] = f32(sigma_scale) * (ma + ms)
    return out


class RGBGridMedium(Medium):
    """RGBGridMedium(; σ_a_grid, σ_s_grid, Le_grid, sigma_scale=1, Le_scale=0, g=0, bounds, transform, majorant_res=(16,16,16))
    (media.jl:1002-1113).  Grids are [nx, ny, nz, 3|4] RGB(A) voxels; an absent σ grid reads as RGBSpectrum(1)."""
    kind = A.HK_MEDIUM_RGB_GRID

    def __init__(self, sigma_a_grid=None, sigma_s_grid=None, Le_grid=None, sigma_scale=1.0, Le_scale=0.0, g=0.0,
                 bounds=((0, 0, 0), (1, 1, 1)), transform=None, majorant_res=(16, 16, 16)):
        assert sigma_a_grid is not None or sigma_s_grid is not None, "At least one of σ_a_grid or σ_s_grid must be provided"
        if Le_grid is not None:
            assert sigma_a_grid is not None, "Le_grid requires σ_a_grid to be provided (following pbrt-v4)"

        def rgba(gd):
            if gd is None:
                return None
            gd = np.asarray(gd, dtype=f32)
            assert gd.ndim == 4 and gd.shape[3] in (3, 4)
            if gd.shape[3] == 3:
                gd = np.concatenate([gd, np.ones(gd.shape[:3] + (1,), f32)], axis=3)
            return np.ascontiguousarray(gd)

        self.sigma_a_grid, self.sigma_s_grid, self.Le_grid = rgba(sigma_a_grid), rgba(sigma_s_grid), rgba(Le_grid)
        shapes = {gd.shape[:3] for gd in (self.sigma_a_grid, self.sigma_s_grid, self.Le_grid) if gd is not None}
        assert len(shapes) == 1, "grids must have the same dimensions"
        self.res = shapes.pop()
        self.sigma_scale, self.Le_scale, self.g = float(f32(sigma_scale)), float(f32(Le_scale)), g
        self.bounds = (tuple(float(f32(v)) for v in bounds[0]), tuple(float(f32(v)) for v in bounds[1]))
        self.medium_to_render = np.eye(4, dtype=f32) if transform is None else np.asarray(transform, dtype=f32)
        self.render_to_medium = np.linalg.inv(self.medium_to_render.astype(np.float64)).astype(f32)
        self.majorant_res = tuple(int(v) for v in majorant_res)
        self.majorant = build_rgb_majorant_grid(self.sigma_a_grid, self.sigma_s_grid, self.sigma_scale, self.res, self.majorant_res)

    def fill_record(self, rec, keep):
        _base_record(rec, self.kind, RGBSpectrum(0.0), RGBSpectrum(0.0), RGBSpectrum(0.0), self.g)
        rec.sigma_scale, rec.Le_scale = self.sigma_scale, self.Le_scale
        rec.bounds_min[:] = self.bounds[0]
        rec.bounds_max[:] = self.bounds[1]
        rec.render_to_medium[:] = [float(x) for x in self.render_to_medium.reshape(-1)]
        rec.medium_to_render[:] = [float(x) for x in self.medium_to_render.reshape(-1)]
        rec.res[:] = self.res
        for name in ("sigma_a_grid", "sigma_s_grid", "Le_grid"):
            gd = getattr(self, name)
            if gd is not None:
                jl = np.ascontiguousarray(np.transpose(gd, (2, 1, 0, 3)))   # C [z][y][x][c] == Julia [x,y,z] of RGBSpectrum
                keep.append(jl)
                setattr(rec, name, jl.ctypes.data_as(A.PF))
        rec.majorant_res[:] = self.majorant_res
        rec.majorant = self.majorant.ctypes.data_as(A.PF)


# ---- NanoVDB -----------------------------------------------------------------------------------------------
LEAF_DIM, LOWER_DIM, UPPER_DIM = 8, 16, 32
LOWER_MASK, UPPER_MASK = 127, 4095
LEAFDATA_SIZE, LEAF_VALUES, LEAF_MASK_OFF, LEAF_MIN_OFF = 2144, 96, 16, 80
LOWER_TABLE, LOWER_CHILDMASK, LOWER_VALUEMASK = 1088, 544, 32
UPPER_TABLE, UPPER_CHILDMASK, UPPER_VALUEMASK = 8256, 4128, 32
UPPER_NODE_SIZE = UPPER_TABLE + 32768 * 8
LOWER_NODE_SIZE = LOWER_TABLE + 4096 * 8
ROOT_HEADER, ROOTTILE_SIZE = 64, 32


def _root_key(c):
    x, y, z = (int(v) & 0xffffffff for v in c)
    return ((z >> 12) & 0x1fffff) | (((y >> 12) & 0x1fffff) << 21) | (((x >> 12) & 0x1fffff) << 42)


def _upper_off(c):
    x, y, z = (int(v) & 0xffffffff for v in c)
    return (((x >> 7) & 31) << 10) | (((y >> 7) & 31) << 5) | ((z >> 7) & 31)


def _lower_off(c):
    x, y, z = (int(v) & 0xffffffff for v in c)
    return (((x >> 3) & 15) << 8) | (((y >> 3) & 15) << 4) | ((z >> 3) & 15)


def build_nanovdb_from_dense(data, origin, extent, background=0.0):
    """nanovdb.jl:602-858.  data[nx, ny, nz] float32 -> (buffer uint8[], metadata dict).  Buffer layout
    [Root | Upper nodes | Lower nodes | Leaf nodes], child offsets relative to the parent, 0-based positions
    here; the `*_offset` metadata are the reference's 1-based byte offsets."""
    data = np.asarray(data, dtype=f32)
    nx, ny, nz = data.shape
    dx, dy, dz = (f32(extent[0]) / nx, f32(extent[1]) / ny, f32(extent[2]) / nz)
    nbx, nby, nbz = -(-nx // 8), -(-ny // 8), -(-nz // 8)
    pad = np.full((nbx * 8, nby * 8, nbz * 8), background, dtype=f32)
    pad[:nx, :ny, :nz] = data
    leaf_coords, leaf_values = [], []
    for bz in range(nbz):
        for by in range(nby):
            for bx in range(nbx):
                blk = pad[bx * 8:bx * 8 + 8, by * 8:by * 8 + 8, bz * 8:bz * 8 + 8]
                if np.any(blk != background):
                    leaf_coords.append((bx * 8, by * 8, bz * 8))
                    leaf_values.append(np.ascontiguousarray(blk).reshape(-1))   # index (lx<<6)|(ly<<3)|lz == C order [lx][ly][lz]
    n_leaves = len(leaf_coords)
    lower_to_leaves = {}
    for li, c in enumerate(leaf_coords):
        lower_to_leaves.setdefault(tuple(v & ~LOWER_MASK for v in c), []).append(li)
    lower_bases = sorted(lower_to_leaves)
    upper_to_lowers = {}
    for low_i, lb in enumerate(lower_bases):
        upper_to_lowers.setdefault(tuple(v & ~UPPER_MASK for v in lb), []).append(low_i)
    upper_bases = sorted(upper_to_lowers)
    n_lowers, n_uppers = len(lower_bases), len(upper_bases)
    root_size = ROOT_HEADER + n_uppers * ROOTTILE_SIZE
    upper_section, lower_section = n_uppers * UPPER_NODE_SIZE, n_lowers * LOWER_NODE_SIZE
    total = root_size + upper_section + lower_section + n_leaves * LEAFDATA_SIZE
    buf = np.zeros(total, dtype=np.uint8)
    upper_pos = lambda i: root_size + i * UPPER_NODE_SIZE
    lower_pos = lambda i: root_size + upper_section + i * LOWER_NODE_SIZE
    order = sorted(range(n_leaves), key=lambda i: leaf_coords[i])
    leaf_pos = [0] * n_leaves
    for slot, li in enumerate(order):
        leaf_pos[li] = root_size + upper_section + lower_section + slot * LEAFDATA_SIZE

    def w32(off, v, dt):
        buf[off:off + 4] = np.array([v], dtype=dt).view(np.uint8)

    def w64(off, v, dt=np.int64):
        buf[off:off + 8] = np.array([v], dtype=dt).view(np.uint8)

    def set_bit(mask_off, n):
        buf[mask_off + (n >> 3)] |= np.uint8(1 << (n & 7))

    for li in range(n_leaves):
        c, vals, off = leaf_coords[li], leaf_values[li], leaf_pos[li]
        for k in range(3):
            w32(off + 4 * k, c[k], np.int32)
        buf[off + 12:off + 15] = 7
        active = vals != background
        bits = np.packbits(active.astype(np.uint8), bitorder="little")
        buf[off + LEAF_MASK_OFF:off + LEAF_MASK_OFF + 64] = bits
        w32(off + LEAF_MIN_OFF, vals.min(), f32)
        w32(off + LEAF_MIN_OFF + 4, vals.max(), f32)
        buf[off + LEAF_VALUES:off + LEAF_VALUES + 2048] = vals.view(np.uint8)
    for low_i, lb in enumerate(lower_bases):
        off = lower_pos(low_i)
        for k in range(3):
            w32(off + 4 * k, lb[k], np.int32)
            w32(off + 12 + 4 * k, lb[k] + 127, np.int32)
        for li in lower_to_leaves[lb]:
            n = _lower_off(leaf_coords[li])
            set_bit(off + LOWER_CHILDMASK, n)
            set_bit(off + LOWER_VALUEMASK, n)
            w64(off + LOWER_TABLE + n * 8, leaf_pos[li] - off)
    for up_i, ub in enumerate(upper_bases):
        off = upper_pos(up_i)
        for k in range(3):
            w32(off + 4 * k, ub[k], np.int32)
            w32(off + 12 + 4 * k, ub[k] + 4095, np.int32)
        for low_i in upper_to_lowers[ub]:
            n = _upper_off(lower_bases[low_i])
            set_bit(off + UPPER_CHILDMASK, n)
            set_bit(off + UPPER_VALUEMASK, n)
            w64(off + UPPER_TABLE + n * 8, lower_pos(low_i) - off)
    cs = np.array(leaf_coords, dtype=np.int64).reshape(-1, 3)
    idx_min = tuple(int(v) for v in cs.min(axis=0)) if n_leaves else (0, 0, 0)
    idx_max = tuple(int(v) + 8 for v in cs.max(axis=0)) if n_leaves else (0, 0, 0)
    for k in range(3):
        w32(4 * k, idx_min[k], np.int32)
        w32(12 + 4 * k, idx_max[k], np.int32)
    w32(24, n_uppers, np.uint32)
    w32(28, background, f32)
    for ti, ub in enumerate(upper_bases):
        t = ROOT_HEADER + ti * ROOTTILE_SIZE
        w64(t, _root_key(ub), np.uint64)
        w64(t + 8, upper_pos(ti))
        w32(t + 16, 1, np.uint32)
        w32(t + 20, background, f32)
    meta = dict(
        world_min=tuple(float(f32(v)) for v in origin),
        world_max=tuple(float(f32(origin[k]) + f32(extent[k])) for k in range(3)),
        inv_mat=(float(f32(1) / dx), 0.0, 0.0, 0.0, float(f32(1) / dy), 0.0, 0.0, 0.0, float(f32(1) / dz)),
        vec=(float(f32(origin[0]) + dx / f32(2)), float(f32(origin[1]) + dy / f32(2)), float(f32(origin[2]) + dz / f32(2))),
        root_offset=1, upper_offset=upper_pos(0) + 1, lower_offset=lower_pos(0) + 1,
        leaf_offset=(leaf_pos[order[0]] + 1) if n_leaves else 1,
        leaf_count=n_leaves, lower_count=n_lowers, upper_count=n_uppers, root_table_size=n_uppers,
        index_min=idx_min, index_max=idx_max, dense=pad, dense_shape=(nx, ny, nz), background=float(background))
    return buf, meta


def build_nanovdb_majorant_grid(meta, bounds, res=(64, 64, 64)):
    """nanovdb.jl:1174-1235: per majorant cell, the two cell corners go through world_to_index_f_raw (full 3x3 inv_mat), the
    integer range is [floor(min - 1), ceil(max + 1)] clipped to [index_min, index_max], and the cell value is the max voxel
    value in that range.  Voxel values come from `meta["dense"]` (index `dense_origin` at [0,0,0]): the array the tree was
    built from, or the tree decoded back to dense (identical to nanovdb_get_value_raw inside it; background outside)."""
    pad = meta["dense"]
    org = meta.get("dense_origin", (0, 0, 0))
    bg = f32(meta["background"])
    inv = [f32(v) for v in meta["inv_mat"]]
    vec = [f32(v) for v in meta["vec"]]
    bmin, bmax = np.array(bounds[0], dtype=f32), np.array(bounds[1], dtype=f32)
    diag = (bmax - bmin).astype(f32)
    rx, ry, rz = res
    out = np.zeros(rx * ry * rz, dtype=f32)
    imin, imax = meta["index_min"], meta["index_max"]
    sx, sy, sz = pad.shape

    def edge(axis, r):   # cell boundaries p_min[i], p_max[i] along one axis, Float32 like the reference
        i = np.arange(r, dtype=f32)
        return (bmin[axis] + diag[axis] * i / f32(r)).astype(f32), (bmin[axis] + diag[axis] * (i + f32(1)) / f32(r)).astype(f32)

    ex, ey, ez = edge(0, rx), edge(1, ry), edge(2, rz)

    def to_index(px, py, pz):
        qx, qy, qz = px - vec[0], py - vec[1], pz - vec[2]
        return (inv[0] * qx + inv[1] * qy + inv[2] * qz, inv[3] * qx + inv[4] * qy + inv[5] * qz, inv[6] * qx + inv[7] * qy + inv[8] * qz)

    if all(inv[k] == 0 for k in (1, 2, 3, 5, 6, 7)):
        # axis-aligned grid (the usual case): a cell's index range factorises per axis, so the box maximum is three successive
        # 1-D range maxima (x, then y, then z) — the same integer ranges and the same voxel values as the general loop below
        def axis_ranges(k, e, n_src):
            a, b = inv[4 * k] * (e[0] - vec[k]), inv[4 * k] * (e[1] - vec[k])
            lo = np.maximum(np.floor(np.minimum(a, b) - f32(1)).astype(np.int64), imin[k])
            hi = np.minimum(np.ceil(np.maximum(a, b) + f32(1)).astype(np.int64), imax[k])
            return lo, hi, np.clip(lo - org[k], 0, n_src - 1), np.clip(hi - org[k], 0, n_src - 1), (lo - org[k] < 0) | (hi - org[k] >= n_src), lo > hi

        def reduce_axis(src, axis, c0, c1):
            out_shape = list(src.shape)
            out_shape[axis] = len(c0)
            dst = np.zeros(out_shape, dtype=f32)
            for i in range(len(c0)):
                sl = [slice(None)] * 3
                sl[axis] = slice(int(c0[i]), int(c1[i]) + 1)
                dl = [slice(None)] * 3
                dl[axis] = i
                if c0[i] <= c1[i]:
                    dst[tuple(dl)] = src[tuple(sl)].max(axis=axis)
            return dst

        rx_ = axis_ranges(0, ex, sx)
        ry_ = axis_ranges(1, ey, sy)
        rz_ = axis_ranges(2, ez, sz)
        m = reduce_axis(reduce_axis(reduce_axis(np.maximum(pad, f32(0)).astype(f32), 0, rx_[2], rx_[3]), 1, ry_[2], ry_[3]), 2, rz_[2], rz_[3])
        outside = rx_[4][:, None, None] | ry_[4][None, :, None] | rz_[4][None, None, :]
        empty = rx_[5][:, None, None] | ry_[5][None, :, None] | rz_[5][None, None, :]
        # a range entirely off the stored array leaves no voxel inside: clip() above would have picked an edge voxel
        off = ((rx_[1] - org[0] < 0) | (rx_[0] - org[0] >= sx))[:, None, None] | ((ry_[1] - org[1] < 0) | (ry_[0] - org[1] >= sy))[None, :, None] | \
              ((rz_[1] - org[2] < 0) | (rz_[0] - org[2] >= sz))[None, None, :]
        m = np.where(off, f32(0), m)
        m = np.where(outside, np.maximum(m, bg), m)
        m = np.where(empty, f32(0), m)
        return np.ascontiguousarray(np.transpose(m, (2, 1, 0))).reshape(-1).astype(f32)      # index x + rx*(y + ry*z)
    for iz in range(rz):
        for iy in range(ry):
            a = to_index(ex[0], ey[0][iy], ez[0][iz])          # vectors over ix
            b = to_index(ex[1], ey[1][iy], ez[1][iz])
            lo = [np.maximum(np.floor(np.minimum(a[k], b[k]) - f32(1)).astype(np.int64), imin[k]) for k in range(3)]
            hi = [np.minimum(np.ceil(np.maximum(a[k], b[k]) + f32(1)).astype(np.int64), imax[k]) for k in range(3)]
            for ix in range(rx):
                x0, x1, y0, y1, z0, z1 = int(lo[0][ix]), int(hi[0][ix]), int(lo[1][ix]), int(hi[1][ix]), int(lo[2][ix]), int(hi[2][ix])
                m = f32(0)
                if x0 <= x1 and y0 <= y1 and z0 <= z1:
                    cx0, cx1 = max(x0 - org[0], 0), min(x1 - org[0], sx - 1)
                    cy0, cy1 = max(y0 - org[1], 0), min(y1 - org[1], sy - 1)
                    cz0, cz1 = max(z0 - org[2], 0), min(z1 - org[2], sz - 1)
                    if cx0 <= cx1 and cy0 <= cy1 and cz0 <= cz1:
                        m = max(m, f32(pad[cx0:cx1 + 1, cy0:cy1 + 1, cz0:cz1 + 1].max()))
                    if x0 - org[0] < 0 or y0 - org[1] < 0 or z0 - org[2] < 0 or x1 - org[0] >= sx or y1 - org[1] >= sy or z1 - org[2] >= sz:
                        m = max(m, bg)
                out[ix + rx * (iy + ry * iz)] = m
    return out


# ---- NanoVDB file IO (nanovdb.jl:868-946 save, 1037-1166 load) ----------------------------------------------------------
GRIDDATA_SIZE, TREEDATA_SIZE = 672, 64
MAP_OFFSET0, WORLDBBOX_OFFSET0 = 296, 560     # 0-based byte offsets of GridData.mMap / mWorldBBox


def save_nanovdb(filepath, buffer, meta):
    """save_nanovdb(filepath, buffer, metadata) (nanovdb.jl:868-946): 736-byte GridData + TreeData header (map, world bbox,
    node offsets/counts), the node data, zlib level 6."""
    import zlib
    header = GRIDDATA_SIZE + TREEDATA_SIZE
    full = np.zeros(header + buffer.size, dtype=np.uint8)
    full[header:] = buffer
    inv = np.array(meta["inv_mat"], dtype=f32)
    m = inv.astype(np.float64)
    cof = np.array([m[4] * m[8] - m[5] * m[7], m[2] * m[7] - m[1] * m[8], m[1] * m[5] - m[2] * m[4],
                    m[5] * m[6] - m[3] * m[8], m[0] * m[8] - m[2] * m[6], m[2] * m[3] - m[0] * m[5],
                    m[3] * m[7] - m[4] * m[6], m[1] * m[6] - m[0] * m[7], m[0] * m[4] - m[1] * m[3]])
    det = m[0] * cof[0] + m[1] * cof[3] + m[2] * cof[6]
    full[MAP_OFFSET0:MAP_OFFSET0 + 36] = (cof / det).astype(f32).view(np.uint8)
    full[MAP_OFFSET0 + 36:MAP_OFFSET0 + 72] = inv.view(np.uint8)
    full[MAP_OFFSET0 + 72:MAP_OFFSET0 + 84] = np.array(meta["vec"], dtype=f32).view(np.uint8)
    full[WORLDBBOX_OFFSET0:WORLDBBOX_OFFSET0 + 48] = np.array(list(meta["world_min"]) + list(meta["world_max"]), dtype=np.float64).view(np.uint8)
    offs = np.array([meta["leaf_offset"] + 63, meta["lower_offset"] + 63, meta["upper_offset"] + 63, meta["root_offset"] + 63], dtype=np.uint64)
    full[GRIDDATA_SIZE:GRIDDATA_SIZE + 32] = offs.view(np.uint8)
    full[GRIDDATA_SIZE + 32:GRIDDATA_SIZE + 44] = np.array([meta["leaf_count"], meta["lower_count"], meta["upper_count"]], dtype=np.uint32).view(np.uint8)
    with open(filepath, "wb") as f:
        f.write(zlib.compress(full.tobytes(), 6))


def extract_nanovdb_metadata(buffer):
    """extract_nanovdb_metadata (nanovdb.jl:1109-1166) of a decompressed FULL buffer (GridData + TreeData + nodes)."""
    wb = buffer[WORLDBBOX_OFFSET0:WORLDBBOX_OFFSET0 + 48].view(np.float64)
    inv = buffer[MAP_OFFSET0 + 36:MAP_OFFSET0 + 72].view(f32)
    vec = buffer[MAP_OFFSET0 + 72:MAP_OFFSET0 + 84].view(f32)
    offs = buffer[GRIDDATA_SIZE:GRIDDATA_SIZE + 32].view(np.uint64)
    counts = buffer[GRIDDATA_SIZE + 32:GRIDDATA_SIZE + 44].view(np.uint32)
    start = GRIDDATA_SIZE + 1
    leaf_off, lower_off, upper_off, root_off = (int(start + int(o)) for o in offs)
    root_table_size = int(buffer[root_off - 1 + 24:root_off - 1 + 28].view(np.uint32)[0])
    background = float(buffer[root_off - 1 + 28:root_off - 1 + 32].view(f32)[0])
    n_leaf = int(counts[0])
    co = np.stack([buffer[leaf_off - 1 + i * LEAFDATA_SIZE:leaf_off - 1 + i * LEAFDATA_SIZE + 12].view(np.int32) for i in range(n_leaf)]) if n_leaf else np.zeros((0, 3), np.int32)
    idx_min = tuple(int(v) for v in co.min(axis=0)) if n_leaf else (0, 0, 0)
    idx_max = tuple(int(v) + LEAF_DIM for v in co.max(axis=0)) if n_leaf else (0, 0, 0)
    return dict(world_min=tuple(float(f32(v)) for v in wb[:3]), world_max=tuple(float(f32(v)) for v in wb[3:]), inv_mat=tuple(float(v) for v in inv),
                vec=tuple(float(v) for v in vec), root_offset=root_off, upper_offset=upper_off, lower_offset=lower_off, leaf_offset=leaf_off,
                leaf_count=n_leaf, lower_count=int(counts[1]), upper_count=int(counts[2]), root_table_size=root_table_size,
                index_min=idx_min, index_max=idx_max, background=background)


def parse_nanovdb_buffer(filepath):
    """parse_nanovdb_buffer (nanovdb.jl:1085-1107): find the zlib stream in the first 500 bytes, inflate, read the metadata."""
    import zlib
    raw = open(filepath, "rb").read()
    start = -1
    for i in range(min(500, len(raw) - 1)):
        if raw[i] == 0x78 and raw[i + 1] in (0x01, 0x5e, 0x9c, 0xda):
            start = i
            break
    if start < 0:
        raise ValueError("Could not find zlib header in NanoVDB file")
    buffer = np.frombuffer(zlib.decompress(raw[start:]), dtype=np.uint8).copy()
    return buffer, extract_nanovdb_metadata(buffer)


def nanovdb_to_dense(buffer, meta):
    """Decode the leaves of a tree back into a dense array over [index_min, index_max) (background elsewhere); the majorant
    builder reads voxels from it.  Non-background tiles are not expected in density grids and are rejected."""
    lo, hi = meta["index_min"], meta["index_max"]
    shape = tuple(max(hi[k] - lo[k], 1) for k in range(3))
    dense = np.full(shape, meta["background"], dtype=f32)
    for i in range(meta["leaf_count"]):
        off = meta["leaf_offset"] - 1 + i * LEAFDATA_SIZE
        c = buffer[off:off + 12].view(np.int32)
        vals = buffer[off + LEAF_VALUES:off + LEAF_VALUES + 2048].view(f32).reshape(8, 8, 8)      # [x][y][z]
        x, y, z = int(c[0]) - lo[0], int(c[1]) - lo[1], int(c[2]) - lo[2]
        dense[x:x + 8, y:y + 8, z:z + 8] = vals
    return dense


class NanoVDBMedium(Medium):
    """NanoVDBMedium(data; bounds, σ_a=0, σ_s=1, g=0, majorant_res=(64,64,64))  (nanovdb.jl:964-1004)."""
    kind = A.HK_MEDIUM_NANOVDB

    def __init__(self, data, bounds, sigma_a=RGBSpectrum(0.0), sigma_s=RGBSpectrum(1.0), g=0.0, majorant_res=(64, 64, 64)):
        self.bounds = (tuple(float(f32(v)) for v in bounds[0]), tuple(float(f32(v)) for v in bounds[1]))
        origin = self.bounds[0]
        extent = tuple(float(f32(self.bounds[1][k]) - f32(self.bounds[0][k])) for k in range(3))
        self.buffer, self.meta = build_nanovdb_from_dense(data, origin, extent)
        self.majorant_res = tuple(int(v) for v in majorant_res)
        self.majorant = build_nanovdb_majorant_grid(self.meta, self.bounds, self.majorant_res)
        self.max_density = float(self.majorant.max())
        self.sigma_a, self.sigma_s, self.g = sigma_a, sigma_s, g

    @classmethod
    def from_file(cls, filepath, sigma_a=RGBSpectrum(0.5), sigma_s=RGBSpectrum(10.0), g=0.0, transform=None, majorant_res=(64, 64, 64)):
        """NanoVDBMedium(filepath; σ_a=0.5, σ_s=10, g=0, transform=I (3x3 medium-to-world), majorant_res=64^3)
        (nanovdb.jl:1320-1422): inv_mat := nanovdb_inv_mat * inv(transform), world bounds = bbox of the transformed corners."""
        self = cls.__new__(cls)
        buffer, meta = parse_nanovdb_buffer(filepath)
        T = np.eye(3, dtype=f32) if transform is None else np.asarray(transform, dtype=f32).reshape(3, 3)
        inv_mat = (np.array(meta["inv_mat"], dtype=f32).reshape(3, 3) @ np.linalg.inv(T.astype(np.float64)).astype(f32)).astype(f32)
        wmin, wmax = meta["world_min"], meta["world_max"]
        corners = np.array([[(wmin, wmax)[i][0], (wmin, wmax)[j][1], (wmin, wmax)[k][2]] for i in (0, 1) for j in (0, 1) for k in (0, 1)], dtype=f32)
        cw = (T @ corners.T).T.astype(f32)
        self.bounds = (tuple(float(v) for v in cw.min(axis=0)), tuple(float(v) for v in cw.max(axis=0)))
        meta = dict(meta, inv_mat=tuple(float(v) for v in inv_mat.reshape(-1)))
        meta["dense"] = nanovdb_to_dense(buffer, meta)
        meta["dense_origin"] = meta["index_min"]
        self.buffer, self.meta = buffer, meta
        self.majorant_res = tuple(int(v) for v in majorant_res)
        self.majorant = build_nanovdb_majorant_grid(meta, self.bounds, self.majorant_res)
        self.max_density = float(self.majorant.max())
        self.sigma_a, self.sigma_s, self.g = sigma_a, sigma_s, g
        return self

    def save(self, filepath):
        """save_nanovdb(filepath, buffer, metadata) of this medium's tree (node-only buffers get the 736-byte header)"""
        if self.meta["root_offset"] != 1:
            raise ValueError("this medium already holds a full file buffer; write self.buffer with zlib instead")
        save_nanovdb(filepath, self.buffer, self.meta)

    def fill_record(self, rec, keep):
        _base_record(rec, self.kind, self.sigma_a, self.sigma_s, RGBSpectrum(0.0), self.g)
        m = self.meta
        rec.bounds_min[:] = self.bounds[0]
        rec.bounds_max[:] = self.bounds[1]
        rec.majorant_res[:] = self.majorant_res
        rec.majorant = self.majorant.ctypes.data_as(A.PF)
        rec.max_density = self.max_density
        rec.nvdb_bytes = self.buffer.ctypes.data_as(C.POINTER(C.c_uint8))
        rec.nvdb_size = self.buffer.size
        rec.root_offset_1based, rec.upper_offset_1based = m["root_offset"], m["upper_offset"]
        rec.lower_offset_1based, rec.leaf_offset_1based = m["lower_offset"], m["leaf_offset"]
        rec.upper_count, rec.lower_count, rec.leaf_count, rec.root_table_size = m["upper_count"], m["lower_count"], m["leaf_count"], m["root_table_size"]
        rec.inv_mat[:] = m["inv_mat"]
        rec.vec[:] = m["vec"]
        rec.index_bbox_min[:] = m["index_min"]
        rec.index_bbox_max[:] = m["index_max"]

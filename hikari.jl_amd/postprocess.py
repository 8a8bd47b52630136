"""postprocess!(film; ...) (src/postprocess.jl:293-357), host side: resolves the keyword arguments into the kernel's parameter
record (tonemap symbol, gamma, FilmSensor -> imaging ratio + Bradford white-balance matrix, background mask) and runs the HIP
kernel through `hk_film_postprocess` / `hk_postprocess`."""
import numpy as np

from . import _abi as A

f32 = np.float32
TONEMAPS = {None: A.HK_TONEMAP_NONE, "reinhard": A.HK_TONEMAP_REINHARD, "reinhard_extended": A.HK_TONEMAP_REINHARD_EXT, "aces": A.HK_TONEMAP_ACES,
            "uncharted2": A.HK_TONEMAP_UNCHARTED2, "filmic": A.HK_TONEMAP_FILMIC}

LMS_FROM_XYZ = np.array([[0.8951, 0.2664, -0.1614], [-0.7502, 1.7135, 0.0367], [0.0389, -0.0685, 1.0296]], dtype=f32)
XYZ_FROM_LMS = np.array([[0.9869929, -0.1470543, 0.1599627], [0.4323053, 0.5183603, 0.0492912], [-0.0085287, 0.0400428, 0.9684867]], dtype=f32)
D65_WHITE_XY = (f32(0.31272), f32(0.32903))


class FilmSensor:
    """FilmSensor(; iso=100, exposure_time=1, white_balance=0) (postprocess.jl:37-47)"""

    def __init__(self, iso=100, exposure_time=1.0, white_balance=0):
        self.iso, self.exposure_time, self.white_balance = f32(iso), f32(exposure_time), f32(white_balance)


def planckian_xy(T):
    """spectral/color.jl:468-493 (CIE 015:2004 approximation), Float32"""
    T = f32(T)
    T2 = T * T
    T3 = T2 * T
    if T <= 4000:
        x = f32(-0.2661239e9) / T3 - f32(0.2343589e6) / T2 + f32(0.8776956e3) / T + f32(0.179910)
    else:
        x = f32(-3.0258469e9) / T3 + f32(2.1070379e6) / T2 + f32(0.2226347e3) / T + f32(0.240390)
    x2 = x * x
    x3 = x2 * x
    if T <= 2222:
        y = f32(-1.1063814) * x3 - f32(1.34811020) * x2 + f32(2.18555832) * x - f32(0.20219683)
    elif T <= 4000:
        y = f32(-0.9549476) * x3 - f32(1.37418593) * x2 + f32(2.09137015) * x - f32(0.16748867)
    else:
        y = f32(3.0817580) * x3 - f32(5.87338670) * x2 + f32(3.75112997) * x - f32(0.37001483)
    return f32(x), f32(y)


def _xy_to_XYZ(x, y):
    return np.array([x / y, f32(1), (f32(1) - x - y) / y], dtype=f32)


def compute_white_balance_matrix(src_temp):
    """Bradford adaptation from a Planckian illuminant at `src_temp` K to D65 (spectral/color.jl:522-547)"""
    src = LMS_FROM_XYZ @ _xy_to_XYZ(*planckian_xy(src_temp))
    dst = LMS_FROM_XYZ @ _xy_to_XYZ(*D65_WHITE_XY)
    scale = np.diag((dst / src).astype(f32)).astype(f32)
    return (XYZ_FROM_LMS @ scale @ LMS_FROM_XYZ).astype(f32)


def make_params(exposure=1.0, tonemap="aces", gamma=2.2, white_point=4.0, sensor=None, background=None):
    if tonemap not in TONEMAPS:
        tonemap = None   # the reference maps any unknown symbol to the linear clamp
    p = A.hk_postprocess_params()
    p.exposure = float(f32(exposure))
    p.tonemap = TONEMAPS[tonemap]
    p.apply_gamma = 0 if gamma is None else 1
    p.inv_gamma = 1.0 if gamma is None else float(f32(1) / f32(gamma))
    p.white_point = float(f32(white_point))
    s = sensor if sensor is not None else FilmSensor()
    p.imaging_ratio = float(s.exposure_time * s.iso / f32(100))
    p.apply_wb = 1 if s.white_balance > 0 else 0
    wb = compute_white_balance_matrix(s.white_balance) if p.apply_wb else np.eye(3, dtype=f32)
    p.wb[:] = [float(v) for v in wb.reshape(-1)]
    p.mask_escaped = 0 if background is None else 1
    p.bg[:] = (0.0, 0.0, 0.0) if background is None else tuple(float(f32(v)) for v in background)
    return p

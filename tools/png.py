"""Tiny PNG writer (zlib only) for eyeballing renders: write_png(path, img[h,w,3] linear HDR, exposure)."""
import struct
import zlib

import numpy as np


def write_png(path, img, exposure=1.0, gamma=2.2):
    a = np.clip(np.nan_to_num(img * exposure), 0, None)
    a = a / (1.0 + a)  # Reinhard
    a = (np.clip(a, 0, 1) ** (1.0 / gamma) * 255 + 0.5).astype(np.uint8)
    h, w, _ = a.shape
    raw = b"".join(b"\x00" + a[y].tobytes() for y in range(h))

    def chunk(t, d):
        c = struct.pack(">I", len(d)) + t + d
        return c + struct.pack(">I", zlib.crc32(t + d) & 0xffffffff)

    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)) +
                chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))

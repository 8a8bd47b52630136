"""Raw-binary fixture sets shared with Julia (julia/make_reference_fixtures.jl): a directory of `<name>.bin` files (little-endian, C /
NumPy order) plus `manifest.txt`, one line per array: `name dtype d0 d1 ...` with dtype in f32 / i32 / u32 / u8.  No NPZ, no JSON: Julia
reads and writes these with `read!` / `write` alone (a row-major [n, k] array is a column-major k x n Matrix there)."""
import os

import numpy as np

DTYPES = {"f32": np.float32, "i32": np.int32, "u32": np.uint32, "u8": np.uint8}
NAMES = {np.dtype(v): k for k, v in DTYPES.items()}


def write_set(directory, arrays):
    os.makedirs(directory, exist_ok=True)
    lines = []
    for name in sorted(arrays):
        a = np.ascontiguousarray(arrays[name])
        assert a.dtype in NAMES, (name, a.dtype)
        a.astype(a.dtype.newbyteorder("<"), copy=False).tofile(os.path.join(directory, name + ".bin"))
        lines.append(" ".join([name, NAMES[a.dtype]] + [str(d) for d in a.shape]))
    with open(os.path.join(directory, "manifest.txt"), "w") as f:
        f.write("\n".join(lines) + "\n")


def read_set(directory):
    """-> {name: array}; {} when the directory holds no manifest"""
    path = os.path.join(directory, "manifest.txt")
    if not os.path.isfile(path):
        return {}
    out = {}
    for line in open(path):
        parts = line.split()
        if not parts:
            continue
        name, dt, shape = parts[0], DTYPES[parts[1]], tuple(int(d) for d in parts[2:])
        a = np.fromfile(os.path.join(directory, name + ".bin"), dtype=np.dtype(dt).newbyteorder("<"))
        out[name] = a.reshape(shape).astype(dt, copy=False)
    return out

#!/usr/bin/env python3
"""Extract the *data tables* embedded in the reference's Julia sources into binary fixtures.

Run once in the build container (needs /root/reference); the outputs are committed:

  hikari.jl_amd/data/sobol_matrices.bin   uint32[1024*52]  (sampler/sobol_matrices.jl:18-6675)
  hikari.jl_amd/data/cie_xyz.bin          float32[3*471]   (spectral/color.jl:53-345, X then Y then Z)
  hikari.jl_amd/data/metal_spectra.bin    see below        (spectral/metal-spectra.jl)

These are numeric tables (pbrt-v4 / CIE data), not code.  SURVEY.md §8c(1) lists them as
the known-answer data the build must share with the reference.
"""
import os, re, struct, sys
import numpy as np

REF = "/root/reference/src"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "hikari.jl_amd", "data")


def grab_block(text, name):
    m = re.search(r"const\s+%s\s*=\s*\w*\[(.*?)\n\]" % re.escape(name), text, re.S)
    if not m:
        raise SystemExit("table %s not found" % name)
    body = re.sub(r"#.*", "", m.group(1))
    return [t for t in re.split(r"[\s,]+", body) if t]


def main():
    os.makedirs(OUT, exist_ok=True)
    sob = open(os.path.join(REF, "sampler/sobol_matrices.jl")).read()
    toks = grab_block(sob, "SobolMatrices32")
    arr = np.array([int(t, 16) for t in toks], dtype=np.uint32)
    assert arr.size == 1024 * 52, arr.size
    arr.tofile(os.path.join(OUT, "sobol_matrices.bin"))

    col = open(os.path.join(REF, "spectral/color.jl")).read()
    xyz = []
    for nm in ("CIE_X", "CIE_Y", "CIE_Z"):
        t = grab_block(col, nm)
        a = np.array([float(x.rstrip("f0")) if x.endswith("f0") else float(x) for x in t], dtype=np.float32)
        assert a.size == 471, (nm, a.size)
        xyz.append(a)
    np.concatenate(xyz).tofile(os.path.join(OUT, "cie_xyz.bin"))

    # measured metal eta/k (pbrt-v4 spectrum data): [name, n, lambdas[n], values[n]] as float32, one record per spectrum
    met = open(os.path.join(REF, "spectral/metal-spectra.jl")).read()
    with open(os.path.join(OUT, "metal_spectra.bin"), "wb") as f:
        names = re.findall(r"const\s+(\w+_(?:ETA|K)_SPECTRUM)\s*=\s*from_interleaved\(PiecewiseLinearSpectrum\{(\d+)\},\s*\((.*?)\)\)", met, re.S)
        f.write(struct.pack("<i", len(names)))
        for name, n, body in names:
            vals = np.array([float(t.rstrip("f0")) if t.endswith("f0") else float(t) for t in re.split(r"[\s,]+", re.sub(r"#.*", "", body)) if t], dtype=np.float32)
            assert vals.size == 2 * int(n), (name, vals.size, n)
            nb = name.encode()
            f.write(struct.pack("<i", len(nb)) + nb + struct.pack("<i", int(n)))
            vals[0::2].tofile(f)
            vals[1::2].tofile(f)

    # Hosek-Wilkie sky model coefficients (c) 2012-2013 Lukas Hosek and Alexander Wilkie, BSD 3-clause (see data/README):
    # 11 bands x 1080 config doubles, then 11 bands x 120 radiance doubles (lights/hosek_wilkie_data.jl:6-1610)
    hw = open(os.path.join(REF, "lights/hosek_wilkie_data.jl")).read()
    bands = [320 + 40 * i for i in range(11)]
    cfg = [np.array([float(t) for t in grab_block(hw, "_HOSEK_SPECTRAL_CONFIG_%d" % b)], dtype=np.float64) for b in bands]
    rad = [np.array([float(t) for t in grab_block(hw, "_HOSEK_SPECTRAL_RAD_%d" % b)], dtype=np.float64) for b in bands]
    assert all(c.size == 1080 for c in cfg) and all(r.size == 120 for r in rad)
    np.concatenate(cfg + rad).tofile(os.path.join(OUT, "hosek_wilkie_sky.bin"))
    # measured medium presets (pbrt-v4 "named media": Jensen et al. 2001, Narasimhan et al. 2006), volpath/media.jl:1769-1829
    med = open(os.path.join(REF, "integrators/volpath/media.jl")).read()
    blk = re.search(r"const _MEDIUM_PRESETS = Dict\{.*?\}\((.*?)\n\)\n", med, re.S).group(1)
    num = lambda t: float(re.sub(r"f(-?\d+)$", r"e\1", t.strip()))
    presets = {}
    for name, ss, sa in re.findall(r'"(\w+)"\s*=>\s*\(σ_s=\(([^)]*)\),\s*σ_a=\(([^)]*)\)\)', blk):
        presets[name] = {"sigma_s": [num(t) for t in ss.split(",")], "sigma_a": [num(t) for t in sa.split(",")]}
    assert len(presets) == 40, len(presets)
    import json
    with open(os.path.join(OUT, "medium_presets.json"), "w") as f:
        json.dump(presets, f, indent=0, sort_keys=True)
    print("wrote", OUT)


if __name__ == "__main__":
    main()

#!/bin/bash
# Print VGPR / scratch / occupancy per kernel of hk_kernels.hip (extra -D flags may be given as arguments).
cd "$(dirname "$0")/../hikari.jl_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -I../../include "$@" -c hk_kernels.hip -o /tmp/hk_regs.o \
    -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c '
import sys, re
cur = None
rows = []
for line in sys.stdin:
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = {"name": m.group(1)}
        rows.append(cur)
        continue
    for key, pat in (("vgpr", r" VGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"), ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"), ("lds", r"LDS Size \[bytes/block\]: (\d+)"), ("sgpr", r"TotalSGPRs: (\d+)")):
        m = re.search(pat, line)
        if m and cur is not None:
            cur[key] = int(m.group(1))
import subprocess
for r in rows:
    name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip().split("(")[0]
    print("%-42s vgpr %3d agpr %3d scratch %4d occ %d lds %6d" % (name[:42], r.get("vgpr", -1), r.get("agpr", -1), r.get("scratch", -1), r.get("occ", -1), r.get("lds", -1)))
'

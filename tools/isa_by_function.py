#!/usr/bin/env python3
"""Static instruction census of one kernel, attributed to the source function each instruction's line belongs to.
    hipcc ... -gline-tables-only -S --cuda-device-only hk_kernels.hip -o k.s ;  isa_by_function.py k.s '_Z7k_shadeILi0E' [csrc dir]
The .loc of an instruction names the innermost inlined source line; functions are found by scanning the sources for definitions that
start at column 0.  Counts are STATIC (loop bodies once): a map of where the code is, not of where the time goes."""
import collections
import os
import re
import sys


def functions(path):
    out = []
    try:
        lines = open(path).read().splitlines()
    except OSError:
        return out
    for i, ln in enumerate(lines, 1):
        m = re.match(r"^(?:template\s*<[^>]*>\s*)?(?:HKD|__global__|__device__|static|inline|extern)\b[^;(]*?\b([A-Za-z_]\w*)\s*\(", ln)
        if m and not ln.startswith(" "):
            out.append((i, m.group(1)))
    return out


def main(asm, prefix, srcdir="hikari.jl_amd/csrc"):
    files, fmap = {}, {}
    text = open(asm).read().splitlines()
    for ln in text:
        m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', ln)
        if m:
            name = m.group(3) or m.group(2)
            files[int(m.group(1))] = os.path.basename(name)
    start = next(i for i, ln in enumerate(text) if ln.startswith(prefix) and ln.rstrip().split(";")[0].strip().endswith(":"))
    cur = ("?", 0)
    by_fn = collections.defaultdict(lambda: collections.Counter())
    for ln in text[start + 1:]:
        s = ln.strip()
        if s.startswith(".loc"):
            p = s.split()
            cur = (files.get(int(p[1]), p[1]), int(p[2]))
            continue
        if s.startswith("s_endpgm"):
            break
        if not s or s.startswith((".", ";")) or s.endswith(":"):
            continue
        op = s.split()[0]
        if op.startswith("v_"):
            cls = "valu"
            if re.match(r"v_(mov|cndmask|cmp|cmpx|readlane|readfirstlane|writelane|accvgpr|swap)", op):
                cls = "valu_move"
        elif op.startswith("s_"):
            cls = "salu"
        elif op.startswith(("global_", "buffer_", "flat_")):
            cls = "vmem"
        elif op.startswith("scratch_"):
            cls = "scratch"
        elif op.startswith("ds_"):
            cls = "lds"
        else:
            cls = "other"
        f, line = cur
        if f not in fmap:
            fmap[f] = functions(os.path.join(srcdir, f))
        name = "?"
        for l0, n in fmap[f]:
            if l0 <= line:
                name = n
            else:
                break
        by_fn[(f, name)][cls] += 1
    tot = collections.Counter()
    for c in by_fn.values():
        tot.update(c)
    print("%-16s %-34s %7s %7s %6s %6s %6s %6s" % ("file", "function", "valu", "v_move", "salu", "vmem", "scr", "lds"))
    for (f, n), c in sorted(by_fn.items(), key=lambda kv: -(kv[1]["valu"] + kv[1]["valu_move"])):
        if c["valu"] + c["valu_move"] + c["vmem"] + c["scratch"] < 8:
            continue
        print("%-16s %-34s %7d %7d %6d %6d %6d %6d" % (f, n, c["valu"], c["valu_move"], c["salu"], c["vmem"], c["scratch"], c["lds"]))
    print("%-16s %-34s %7d %7d %6d %6d %6d %6d" % ("total", "", tot["valu"], tot["valu_move"], tot["salu"], tot["vmem"], tot["scratch"], tot["lds"]))


if __name__ == "__main__":
    main(*sys.argv[1:4])

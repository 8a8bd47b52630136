#!/usr/bin/env python3
"""Loops of one kernel's disassembly (a file written by tools/isa_census.py --dump): every backward branch closes a loop
[target, branch]; prints per loop its static VALU / SALU / SMEM / VMEM / branch / waitcnt+nop counts (nested loops included in their
parents) and the transcendental ops inside, so that the phases of a state machine can be told apart.
    tools/isa_loops.py /tmp/isa/k/void_k_track_flat_8_true_.s [--show A B]     (--show: print the instructions of lines A..B)"""
import collections
import re
import sys


def parse(path):
    ins = []
    for ln in open(path).read().splitlines():
        m = re.match(r"^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):", ln)
        if m:
            ins.append((int(m.group(3), 16), m.group(1), m.group(2)))
    return ins


def cls(op):
    if op.startswith("v_"):
        return "valu"
    if op.startswith(("s_cbranch", "s_branch")):
        return "branch"
    if op.startswith(("s_waitcnt", "s_nop")):
        return "wait"
    if op.startswith(("s_load", "s_buffer_load")):
        return "smem"
    if op.startswith("s_"):
        return "salu"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("ds_"):
        return "lds"
    return "other"


def main(argv):
    path = argv[0]
    ins = parse(path)
    addr_idx = {a: i for i, (a, _, _) in enumerate(ins)}
    if "--show" in argv:
        i = argv.index("--show")
        a, b = int(argv[i + 1]), int(argv[i + 2])
        for k in range(a, min(b + 1, len(ins))):
            print("%5d  %-28s %s" % (k, ins[k][1], ins[k][2]))
        return
    loops = []
    for i, (a, op, args) in enumerate(ins):
        if op.startswith(("s_cbranch", "s_branch")):
            m = re.search(r"(\d+)\s*$", args) or re.search(r"<\S+\+0x([0-9a-fA-F]+)>", args)
            # llvm-objdump prints the target as a signed word offset: s_cbranch_x <offset>
            try:
                off = int(args.split()[-1])
            except ValueError:
                continue
            if off >= 32768:
                off -= 65536
            tgt = a + 4 + 4 * off
            if tgt <= a and tgt in addr_idx:
                loops.append((addr_idx[tgt], i))
    loops.sort()
    for s, e in loops:
        c = collections.Counter(cls(op) for _, op, _ in ins[s:e + 1])
        tr = collections.Counter(op.split("_e")[0] for _, op, _ in ins[s:e + 1] if re.match(r"v_(log|exp|rcp|sqrt|rsq|div_scale|div_fmas|div_fixup|floor|cndmask|readlane|writelane|mad_u64)", op))
        print("loop ins %5d..%5d  (%4d)  %s   %s" % (s, e, e - s + 1, "  ".join("%s %d" % kv for kv in sorted(c.items())), dict(tr)))


if __name__ == "__main__":
    main(sys.argv[1:])

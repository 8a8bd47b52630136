#!/usr/bin/env python3
"""Static instruction census of kernels in the BUILT library object (no recompilation):
    tools/isa_census.py <kernel-name-substring> [...]        e.g.  tools/isa_census.py k_track_flat 'k_trackILi8ELb1'
Extracts the gfx950 code object from hk_kernels.o's .hip_fatbin, disassembles it once (/tmp/isa/dev.s) and prints per matching kernel:
VGPRs / SGPRs / scratch / LDS from the kernel descriptor notes and the static count of VALU / SALU / VMEM / LDS / branch instructions.
--dump <dir> also writes each kernel's disassembly there."""
import collections
import os
import re
import struct
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJ = os.path.join(ROOT, "hikari.jl_amd", "csrc", "hk_kernels.o")
TMP = "/tmp/isa"
LLVM = "/opt/rocm/lib/llvm/bin"


def code_object():
    os.makedirs(TMP, exist_ok=True)
    fat, co, dis = TMP + "/fat.bin", TMP + "/dev.co", TMP + "/dev.s"
    if os.path.exists(dis) and os.path.getmtime(dis) > os.path.getmtime(OBJ):
        return co, dis
    subprocess.run(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", OBJ, fat], check=True)
    b = open(fat, "rb").read()
    assert b[:24] == b"__CLANG_OFFLOAD_BUNDLE__"
    n = struct.unpack_from("<Q", b, 24)[0]
    pos = 32
    for _ in range(n):
        off, size, idl = struct.unpack_from("<QQQ", b, pos)
        pos += 24
        tid = b[pos:pos + idl].decode()
        pos += idl
        if "gfx950" in tid:
            open(co, "wb").write(b[off:off + size])
    with open(dis, "w") as f:
        subprocess.run([LLVM + "/llvm-objdump", "-d", "--no-show-raw-insn", co], stdout=f, check=True)
    return co, dis


def notes(co):
    out = subprocess.run([LLVM + "/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
    res = {}
    cur = {}
    for ln in out.splitlines():
        m = re.match(r"\s*-?\s*\.(\w+):\s*(.*)", ln)
        if not m:
            continue
        k, v = m.group(1), m.group(2).strip()
        if k == "name":
            cur = res.setdefault(v.strip("'\""), cur if "name" not in cur else {})
            cur["name"] = v
        cur[k] = v
    return out


def main(argv):
    dump = None
    if "--dump" in argv:
        i = argv.index("--dump")
        dump = argv[i + 1]
        argv = argv[:i] + argv[i + 2:]
        os.makedirs(dump, exist_ok=True)
    co, dis = code_object()
    meta = subprocess.run([LLVM + "/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
    # crude YAML scan: blocks separated by "- .agpr_count" ... collect per .name
    blocks = re.split(r"\n\s*- \.agpr_count", meta)
    info = {}
    for b in blocks:
        m = re.search(r"\.name:\s+(\S+)", b)
        if not m:
            continue
        g = lambda k: (re.search(r"\.%s:\s+(\d+)" % k, b) or [None, "?"])[1]
        info[m.group(1).strip("'\"")] = dict(vgpr=g("vgpr_count"), sgpr=g("sgpr_count"), scratch=g("private_segment_fixed_size"), lds=g("group_segment_fixed_size"),
                                             vspill=g("vgpr_spill_count"), sspill=g("sgpr_spill_count"))
    text = open(dis).read().splitlines()
    starts = [(i, re.match(r"^[0-9a-f]+ <(\S+)>:", ln).group(1)) for i, ln in enumerate(text) if re.match(r"^[0-9a-f]+ <\S+>:", ln)]
    for pat in argv:
        for idx, (i, name) in enumerate(starts):
            if pat not in name:
                continue
            end = starts[idx + 1][0] if idx + 1 < len(starts) else len(text)
            c = collections.Counter()
            for ln in text[i + 1:end]:
                t = ln.strip()
                if not t or t.startswith("<") or t.endswith(":"):
                    continue
                op = t.split()[0]
                if op.startswith("v_"):
                    c["valu"] += 1
                    if op.startswith(("v_readlane", "v_writelane", "v_readfirstlane")):
                        c["v_lane"] += 1
                    if op.startswith("v_cndmask"):
                        c["v_cndmask"] += 1
                elif op.startswith("s_"):
                    if op.startswith(("s_cbranch", "s_branch")):
                        c["branch"] += 1
                    elif op.startswith("s_waitcnt"):
                        c["waitcnt"] += 1
                    elif op.startswith("s_nop"):
                        c["nop"] += 1
                    elif op.startswith("s_load") or op.startswith("s_buffer_load"):
                        c["smem"] += 1
                    else:
                        c["salu"] += 1
                elif op.startswith(("global_", "buffer_", "flat_", "scratch_")):
                    c["vmem"] += 1
                    if op.startswith("scratch_"):
                        c["scratch_ops"] += 1
                elif op.startswith("ds_"):
                    c["lds"] += 1
                else:
                    c["other"] += 1
            dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip().split("(")[0]
            print("%-60s %s" % (dem[:60], info.get(name, {})))
            print("    " + "  ".join("%s %d" % kv for kv in sorted(c.items())))
            if dump:
                open(os.path.join(dump, re.sub(r"\W+", "_", dem)[:80] + ".s"), "w").write("\n".join(text[i:end]))


if __name__ == "__main__":
    main(sys.argv[1:])

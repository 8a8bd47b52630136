#!/usr/bin/env python3
"""Per-bounce kernel time of the LAST pass in a rocprofv3 rocpd database (--kernel-trace): the dispatches after the last k_camera are
split at every closest-hit launch (k_trace / k_trace_lean: one per depth) and summed per kernel family — shows where a frame's time sits
along the depth axis (the full-size bounces, the thinning middle, the per-launch floor of the tail).   rocpd_by_depth.py <db>"""
import collections
import re
import sqlite3
import sys


def family(name):
    m = re.search(r"(k_[a-z_]+)", name)
    return m.group(1) if m else name[:24]


def main(path):
    db = sqlite3.connect(path)
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    s, e = ("start", "end") if "start" in cols else ("start_time", "end_time")
    rows = db.execute("select name, %s, %s from kernels order by %s" % (s, e, s)).fetchall()
    cams = [i for i, r in enumerate(rows) if "k_camera" in r[0]]
    rows = rows[cams[-1]:]
    depth, per, fams = -1, [], []
    for name, a, b in rows:
        f = family(name)
        if f in ("k_trace", "k_trace_lean"):
            depth += 1
        d = max(depth, 0)
        while len(per) <= d:
            per.append(collections.Counter())
        per[d][f] += (b - a) / 1e3
        per[d]["_n"] += 1
        if f not in fams:
            fams.append(f)
    span = (max(r[2] for r in rows) - rows[0][1]) / 1e3
    print("last pass: %d dispatches, span %.1f us, kernels %.1f us" % (len(rows), span, sum(sum(v for k, v in c.items() if k != "_n") for c in per)))
    print("depth  n  total_us  " + "  ".join("%s" % f[2:14].rjust(12) for f in fams))
    cum = 0.0
    for d, c in enumerate(per):
        tot = sum(v for k, v in c.items() if k != "_n")
        cum += tot
        print("%5d %2d %9.1f  " % (d, c["_n"], tot) + "  ".join("%12.1f" % c.get(f, 0.0) for f in fams) + "   cum %.0f" % cum)


if __name__ == "__main__":
    main(sys.argv[1])

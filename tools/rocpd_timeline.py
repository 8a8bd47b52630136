#!/usr/bin/env python3
"""The LAST n kernel dispatches of a rocprofv3 rocpd database (--kernel-trace) in start order: name, start offset, duration and the gap
to the previous kernel's end — the timeline of one small render call (tools/gpu_prog.sh).   rocpd_timeline.py <db> [n]"""
import sqlite3
import sys


def main(path, n):
    db = sqlite3.connect(path)
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    s, e = ("start", "end") if "start" in cols else ("start_time", "end_time")
    rows = db.execute("select name, %s, %s, grid_x, workgroup_x from kernels order by %s" % (s, e, s)).fetchall()[-n:]
    t0, prev_end = rows[0][1], rows[0][1]
    busy = 0.0
    for name, a, b, g, wg in rows:
        nm = name.replace("void ", "")
        nm = nm if len(nm) <= 48 else nm[:45] + "..."
        print("%-48s start %9.1f us  dur %8.1f us  gap %7.1f us  grid %d/%d" % (nm, (a - t0) / 1e3, (b - a) / 1e3, (a - prev_end) / 1e3, g, wg))
        busy += (b - a) / 1e3
        prev_end = max(prev_end, b)
    print("span %.1f us, kernels %.1f us" % ((prev_end - t0) / 1e3, busy))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 60)

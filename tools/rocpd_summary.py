#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (--kernel-trace [--stats] [--pmc ...]) into the text tables committed
under profiles/: per-kernel calls / total / average / min / max duration, registers, and PMC counter sums."""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    rows = cur.execute("""select name, count(*), sum(duration), avg(duration), min(duration), max(duration),
                                 max(vgpr_count), max(sgpr_count), max(lds_size), max(scratch_size), max(grid_x), max(workgroup_x)
                          from kernels group by name order by sum(duration) desc""").fetchall()
    total = sum(r[2] for r in rows) or 1
    print("%-60s %7s %12s %11s %11s %11s %6s %5s %5s %7s %8s %9s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "pct", "vgpr", "sgpr", "lds_B", "scratch", "grid/wg"))
    for r in rows:
        name = r[0] if len(r[0]) <= 60 else r[0][:57] + "..."
        print("%-60s %7d %12.1f %11.2f %11.2f %11.2f %6.2f %5d %5d %7d %8d %5d/%-3d" % (name, r[1], r[2] / 1e3, r[3] / 1e3, r[4] / 1e3, r[5] / 1e3,
                                                                                 100.0 * r[2] / total, r[6], r[7], r[8], r[9], r[10], r[11]))
    try:
        pmc = cur.execute("""select name, counter_name, count(*), sum(counter_value) from pmc_events
                             group by name, counter_name order by name""").fetchall()
    except sqlite3.Error:
        pmc = []
    if pmc:
        print("\n%-60s %-24s %8s %18s %18s" % ("kernel", "counter", "launches", "sum", "per_launch"))
        for name, cn, n, v in pmc:
            name = name if len(name) <= 60 else name[:57] + "..."
            print("%-60s %-24s %8d %18.1f %18.1f" % (name, cn, n, v, v / max(n, 1)))


if __name__ == "__main__":
    main(sys.argv[1])

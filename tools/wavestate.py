#!/usr/bin/env python3
"""Per kernel: the share of resident wave time parked at s_waitcnt / barriers (SQ_WAIT_ANY), stalled at issue (SQ_WAIT_INST_ANY) and
issuing (SQ_ACTIVE_INST_ANY), each over SQ_WAVE_CYCLES, from two rocpd_summary.py tables (tools/gpu_wavestate.sh).
    wavestate.py <wait table> <active table>"""
import re
import sys


def counters(path):
    out = {}
    on = False
    for ln in open(path).read().splitlines():
        if ln.startswith("kernel") and "counter" in ln:
            on = True
            continue
        if not on or not ln.strip():
            continue
        m = re.match(r"(.{60})\s+(\S+)\s+(\d+)\s+([\d.]+)", ln)
        if m:
            out[(m.group(1).strip(), m.group(2))] = float(m.group(4))
    return out


c = counters(sys.argv[1])
c.update(counters(sys.argv[2]))
names = sorted({k for k, _ in c}, key=lambda k: -c.get((k, "SQ_WAVE_CYCLES"), 0.0))
print("%-60s %14s  %7s %7s %7s" % ("kernel", "wave_cycles", "parked", "stalled", "issuing"))
for k in names:
    wc = c.get((k, "SQ_WAVE_CYCLES"), 0.0)
    if wc <= 0:
        continue
    print("%-60s %14.4g  %7.3f %7.3f %7.3f" % (k, wc, c.get((k, "SQ_WAIT_ANY"), 0) / wc, c.get((k, "SQ_WAIT_INST_ANY"), 0) / wc, c.get((k, "SQ_ACTIVE_INST_ANY"), 0) / wc))

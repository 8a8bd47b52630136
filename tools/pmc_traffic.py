#!/usr/bin/env python3
"""profiles/pmc_traffic_<config>.json from three rocprofv3 rocpd databases (separate --pmc passes of the same bench command):
    pmc_traffic.py FETCH.db WRITE.db L2.db "<description of the command>" > profiles/pmc_traffic_<config>.json
Per kernel family (k_trace*, k_shade*, k_shadow*, k_camera, k_film, k_track*, k_scatter): FETCH_SIZE / WRITE_SIZE averages per
launch in KB and hbm_bytes_per_launch = (2*FETCH_SIZE + WRITE_SIZE) * 1024 — FETCH_SIZE doubled per
/opt/skills/guides/MI355X_MICROARCH.md (gfx950 tallies the 128-B requests of these 16 B/lane reads at 64 B); raw kept alongside."""
import json
import sqlite3
import sys

FAMILIES = ["k_trace", "k_shade", "k_shadow", "k_camera", "k_film", "k_track", "k_scatter", "k_escaped", "k_light_select", "k_walk"]


def family(name):
    n = name.replace("void ", "")
    for f in sorted(FAMILIES, key=len, reverse=True):
        if n.startswith(f):
            return f
    return None


def per_launch(path, counters):
    db = sqlite3.connect(path)
    rows = db.execute("select name, counter_name, dispatch_id, sum(counter_value) from pmc_events group by name, counter_name, dispatch_id").fetchall()
    acc = {}
    for name, cn, _, v in rows:
        f = family(name)
        if f is None or cn not in counters:
            continue
        a = acc.setdefault((f, cn), [0.0, 0])
        a[0] += v
        a[1] += 1
    return {k: v[0] / max(v[1], 1) for k, v in acc.items()}, {k[0]: v[1] for k, v in acc.items()}


def main(fetch_db, write_db, l2_db, what):
    fe, n_f = per_launch(fetch_db, ("FETCH_SIZE",))
    wr, _ = per_launch(write_db, ("WRITE_SIZE",))
    l2, _ = per_launch(l2_db, ("TCC_HIT_sum", "TCC_MISS_sum"))
    out = {"_comment": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / TCC_HIT_sum TCC_MISS_sum (separate passes), averages per launch; "
                       "hbm_bytes_per_launch = (2*FETCH_SIZE + WRITE_SIZE)*1024 (FETCH_SIZE doubled per MI355X_MICROARCH.md, raw kept alongside). " + what}
    for f in FAMILIES:
        if (f, "FETCH_SIZE") not in fe:
            continue
        fk, wk = fe[(f, "FETCH_SIZE")], wr.get((f, "WRITE_SIZE"), 0.0)
        e = {"launches_profiled": n_f[f], "fetch_kb": round(fk, 1), "write_kb": round(wk, 1), "hbm_bytes_per_launch": int((2 * fk + wk) * 1024),
             "raw_bytes_per_launch": int((fk + wk) * 1024)}
        h, m = l2.get((f, "TCC_HIT_sum")), l2.get((f, "TCC_MISS_sum"))
        if h is not None and m is not None and h + m > 0:
            e["l2_hit_rate"] = round(h / (h + m), 3)
        out[f] = e
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(*sys.argv[1:5])

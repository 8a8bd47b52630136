#!/usr/bin/env python3
"""profiles/utilisation_<config>.json from the SQ / GRBM counter summaries of tools/profile_round.sh (rocpd_summary.py text tables):
    pmc_utilisation.py <dir> <round> <config> > profiles/<round>_utilisation_<config>.json
Per kernel family (k_trace*, k_shade*, k_shadow*, k_track*, k_scatter, k_camera, k_film, k_escaped), per launch:
    avg_launch_us          rocprofv3 kernel trace
    valu_inst_per_launch   SQ_INSTS_VALU (wave-instructions)
    valu_issue_frac        SQ_INSTS_VALU * 2 cycles / (GRBM_GUI_ACTIVE * 1024 SIMDs): a wave64 VALU instruction occupies a SIMD-32 for 2
                           cycles (MI355X_MICROARCH.md); quarter-rate instructions (v_rcp / v_sqrt / v_exp, 32-bit integer
                           multiplies) are counted at the same 2 cycles, so this is a LOWER bound of the issue-port occupancy
    valu_busy              SQ_ACTIVE_INST_VALU * 4 / (GRBM_GUI_ACTIVE * 1024 SIMDs): rocprof's own derived metric VALUBusy — the share of the SIMD
                           cycles in which the vector ALU was executing (the SQ counts in units of 4 cycles; SQ_WAVE_CYCLES * 4 over the
                           same denominator = resident waves per SIMD comes out at 3.8 for the kernels compiled for 4: the unit is right).
                           The MEASURED pipe occupancy: ~4.1 cycles per issued wave64 instruction in every kernel here (tools/valu_rate.hip: 4.3 -
                           4.5 for most opcodes, 2.3 - 2.9 for v_fma / v_mul / v_mov, 8.2 for the transcendentals)
    waves_per_simd         SQ_WAVE_CYCLES * 4 / (GRBM_GUI_ACTIVE * 1024 SIMDs): resident waves per SIMD, averaged over the launch
    lane_util              SQ_THREAD_CYCLES_VALU / (64 * SQ_ACTIVE_INST_VALU): active lanes per issued VALU instruction
    wait_frac              SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES: share of resident wave time spent waiting for any instruction's operands
    vmem_rd_per_launch     SQ_INSTS_VMEM_RD, lds_per_launch SQ_INSTS_LDS
"""
import json
import re
import sys

FAMILIES = ["k_trace", "k_shade", "k_shadow", "k_camera", "k_film", "k_track", "k_scatter", "k_escaped", "k_light_select", "k_walk"]


def family(name):
    n = name.replace("void ", "")
    for f in sorted(FAMILIES, key=len, reverse=True):
        if n.startswith(f):
            return f
    return None


def parse(path):
    """-> ({family: (launches, total_us)}, {(family, counter): sum})"""
    t, c = {}, {}
    try:
        lines = open(path).read().splitlines()
    except OSError:
        return t, c
    in_pmc = False
    for ln in lines:
        if ln.startswith("kernel") and "counter" in ln:
            in_pmc = True
            continue
        if not ln.strip() or ln.startswith("kernel"):
            continue
        if not in_pmc:
            m = re.match(r"(.{60})\s+(\d+)\s+([\d.]+)\s+([\d.]+)", ln)
            if m and family(m.group(1).strip()):
                f = family(m.group(1).strip())
                a = t.setdefault(f, [0, 0.0])
                a[0] += int(m.group(2))
                a[1] += float(m.group(3))
        else:
            m = re.match(r"(.{60})\s+(\S+)\s+(\d+)\s+([\d.]+)", ln)
            if m and family(m.group(1).strip()):
                key = (family(m.group(1).strip()), m.group(2))
                c[key] = c.get(key, 0.0) + float(m.group(4))
    return t, c


def main(d, rnd, cfg):
    out = {"_comment": "SQ / GRBM counters (rocprofv3 --pmc, separate passes of bench.py --config %s), per launch; see tools/pmc_utilisation.py for the definitions. round %s." % (cfg, rnd)}
    sets = {}
    for name in ("sq_valu", "sq_busy", "sq_mem", "grbm"):
        sets[name] = parse("%s/%s_%s_%s.txt" % (d, rnd, name, cfg))
    timing, _ = sets["sq_valu"]
    counters = {}
    for name in sets:
        counters.update(sets[name][1])
    for f in FAMILIES:
        if f not in timing:
            continue
        n, tot = timing[f]
        g = lambda k: counters.get((f, k))
        e = {"launches_profiled": n, "avg_launch_us": round(tot / n, 2)}
        iv, tc, ai, gui, wc, wi = g("SQ_INSTS_VALU"), g("SQ_THREAD_CYCLES_VALU"), g("SQ_ACTIVE_INST_VALU"), g("GRBM_GUI_ACTIVE"), g("SQ_WAVE_CYCLES"), g("SQ_WAIT_INST_ANY")
        if iv is not None:
            e["valu_inst_per_launch"] = int(iv / n)
        if iv is not None and gui:
            e["valu_issue_frac"] = round(iv * 2.0 / (gui / 8.0 * 1024.0), 4)     # GRBM_GUI_ACTIVE is reported once per XCD (8): /8 = kernel cycles
        if ai is not None and gui:
            e["valu_busy"] = round(ai * 4.0 / (gui / 8.0 * 1024.0), 4)
        if wc and gui:
            e["waves_per_simd"] = round(wc * 4.0 / (gui / 8.0 * 1024.0), 3)
        if tc is not None and ai:
            e["lane_util"] = round(tc / (64.0 * ai), 4)
        if wc and wi is not None:
            e["wait_frac"] = round(wi / wc, 4)
        for k, label in (("SQ_INSTS_VMEM_RD", "vmem_rd_per_launch"), ("SQ_INSTS_LDS", "lds_per_launch"), ("SQ_INSTS_SALU", "salu_per_launch")):
            if g(k) is not None:
                e[label] = int(g(k) / n)
        out[f] = e
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(*sys.argv[1:4])

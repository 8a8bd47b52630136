"""Achieved-accuracy bookkeeping of the sub-kernel parity tests (the loose `rtol` of those tests said nothing
about what the device achieves, so a 100-ulp regression would have passed).  Every call prints the achieved maximum in units of the
binary32 spacing at the reference value and asserts it against tests/golden/ulp_bounds.json, which holds 2 x the value measured on
the MI355X when the bound was recorded (never less than 2).  `HK_RECORD_ULP=1 pytest -m gpu ...` writes what it measures to
gpurun_out/ulp_achieved.json instead of asserting (tools/ulp_bounds_update.py turns that file into the new bounds).
Kinds above 2 ulp are listed, with the reason, in DESIGN.md §2."""
import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BOUNDS_FILE = os.path.join(ROOT, "tests", "golden", "ulp_bounds.json")
RECORD_FILE = os.path.join(ROOT, "gpurun_out", "ulp_achieved.json")


def ulp_eff(out, ref, floor):
    """|out - ref| in spacings of binary32 at max(|ref|, floor): plain ulps away from zero, an absolute measure below `floor`
    (where a relative one would only measure cancellation noise)."""
    out, ref = np.asarray(out, np.float32), np.asarray(ref, np.float32)
    scale = np.spacing(np.maximum(np.abs(ref), np.float32(floor)).astype(np.float32)).astype(np.float64)
    return np.abs(out.astype(np.float64) - ref.astype(np.float64)) / scale


def _bounds():
    try:
        return json.load(open(BOUNDS_FILE))
    except (OSError, ValueError):
        return {}


def check(key, out, ref, floor, rows=None):
    """max effective ulp of out vs ref (over `rows` when given): printed, and asserted against the recorded bound."""
    e = ulp_eff(out, ref, floor)
    if rows is not None:
        e = e[rows]
    achieved = float(e.max()) if e.size else 0.0
    print("achieved[%s] = %.1f ulp" % (key, achieved))
    if os.environ.get("HK_RECORD_ULP") == "1":
        os.makedirs(os.path.dirname(RECORD_FILE), exist_ok=True)
        try:
            rec = json.load(open(RECORD_FILE))
        except (OSError, ValueError):
            rec = {}
        rec[key] = max(achieved, rec.get(key, 0.0))
        json.dump(rec, open(RECORD_FILE, "w"), indent=1, sort_keys=True)
        return achieved
    b = _bounds()
    assert key in b, "no recorded bound for %s (run with HK_RECORD_ULP=1 on the GPU box, then tools/ulp_bounds_update.py)" % key
    assert achieved <= b[key], "%s: %.1f ulp achieved, bound %.1f" % (key, achieved, b[key])
    return achieved

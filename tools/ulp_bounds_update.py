#!/usr/bin/env python3
"""gpurun_out/ulp_achieved.json (written by `HK_RECORD_ULP=1 pytest -m gpu`) -> tests/golden/ulp_bounds.json: bound = max(2, 2 x achieved),
rounded up to a whole ulp.  Prints the kinds above 2 ulp (DESIGN.md §2 explains each)."""
import json
import math
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ach = json.load(open(os.path.join(ROOT, "gpurun_out", "ulp_achieved.json")))
bounds = {k: float(max(2, math.ceil(2.0 * v))) for k, v in sorted(ach.items())}
json.dump({"_comment": "max |device - oracle| in binary32 spacings (tests/ulp_bounds.py): 2 x the value achieved on the MI355X when recorded, at least 2",
           **bounds}, open(os.path.join(ROOT, "tests", "golden", "ulp_bounds.json"), "w"), indent=1)
for k, v in sorted(ach.items()):
    print("%-60s achieved %8.1f  bound %8.0f %s" % (k, v, bounds[k], "" if v <= 2 else "  > 2 ulp"))

#!/usr/bin/env python3
"""Dev tool: per-slot average launch times of bench lines side by side.  usage: show_kernels.py a.json b.json ..."""
import json, sys
runs = [json.loads(open(f).read().strip().split("\n")[-1]) for f in sys.argv[1:]]
names = sorted({k for r in runs for k in r.get("kernels", {})})
print("%-22s" % "ms_per_step" + "".join("%12.4f" % r["ms_per_step"] for r in runs))
for n in names:
    print("%-22s" % n + "".join("%12.2f" % r["kernels"].get(n, {}).get("avg_us", float("nan")) for r in runs))
for ph in ("u_step", "v_step"):
    print("%-22s" % ("wall " + ph) + "".join("%12.1f" % r.get("roofline_phase", {}).get(ph, {}).get("wall_us_per_step", float("nan")) for r in runs))

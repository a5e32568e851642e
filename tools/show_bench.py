#!/usr/bin/env python3
"""Dev tool: per-kernel table of a bench.py JSON line."""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
for k, v in sorted(d["kernels"].items(), key=lambda kv: -kv[1]["gpu_time_share"]):
    print(f"{k:18s} avg_us {v['avg_us']:9.2f} timed {v['timed_launches']:4d} share {v['gpu_time_share']:.4f} alg {v['algorithmic_bytes']/1e6:8.2f} MB "
          f"frac {v['frac_hbm_peak']:.4f} traffic {v['traffic_bytes']}")
print("ms_per_step", d["ms_per_step"], "value", d["value"], "u_step wall us", (d.get("roofline_phase") or {}).get("u_step", {}).get("wall_us_per_step"))
print("roofline", d["roofline"])

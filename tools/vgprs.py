#!/usr/bin/env python3
"""Dev tool: VGPR / spill / occupancy per kernel from the device assembly (hipcc --cuda-device-only -S)."""
import re, subprocess, sys
txt = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else ""
rows = []
for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", txt, re.S):
    b = m.group(2)
    g = lambda k: int(re.search(r"\.amdhsa_%s (\d+)" % k, b).group(1))
    rows.append((m.group(1), g("next_free_vgpr"), g("accum_offset"), g("group_segment_fixed_size"), g("private_segment_fixed_size")))
names = subprocess.run(["c++filt"] + [r[0] for r in rows], capture_output=True, text=True).stdout.strip().split("\n")
for r, n in zip(rows, names):
    n = n.replace("void ", "")
    if pat in n:
        print(f"vgpr+agpr {r[1]:4d} (arch {r[2]:3d})  scratch {r[4]:5d}  waves/SIMD {min(8, 512 // max(r[1], 1)):d}  {n[:100]}")

#!/usr/bin/env python3
"""Dev tool: achieved occupancy per kernel from one rocprofv3 --pmc pass (SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES
GRBM_GUI_ACTIVE).  SQ_WAVE_CYCLES counts quad-cycles summed over all waves (MI355X_MICROARCH.md); GRBM_GUI_ACTIVE the
busy cycles of the dispatch SUMMED OVER THE 8 XCDs -- calibrated on the solver's own one-wave k_spin kernel of the same
run (one wave in flight for 287 us: GRBM_GUI_ACTIVE = 8.5 x 4 x SQ_WAVE_CYCLES).  Mean waves in flight of a kernel =
4 * SQ_WAVE_CYCLES / (GRBM_GUI_ACTIVE / that factor), against 256 CUs x 32 wave slots = 8192.
Usage: pmc_occupancy.py <dir> <out.txt>"""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/*/*counter_collection.csv")[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
seen = set()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    key = (r.get("Dispatch_Id"), k)
    if key not in seen: seen.add(key); n[k] += 1
rows = []
spin = [c for k, c in acc.items() if k.startswith("k_spin")]
xcd = spin[0]["GRBM_GUI_ACTIVE"] / (4.0 * spin[0]["SQ_WAVE_CYCLES"]) if spin and spin[0].get("SQ_WAVE_CYCLES") else 8.0
for k, c in acc.items():
    gui = c.get("GRBM_GUI_ACTIVE", 0.0) / xcd
    if gui <= 0 or k.startswith("k_spin") or k.startswith("__amd"): continue
    inflight = 4.0 * c.get("SQ_WAVE_CYCLES", 0.0) / gui
    rows.append((c.get("SQ_WAVE_CYCLES", 0.0), k, n[k], c.get("SQ_WAVES", 0.0) / max(n[k], 1), inflight, inflight / 8192.0,
                 c.get("VGPR_Count", 0)))
rows.sort(reverse=True)
out = [f"(GRBM_GUI_ACTIVE / {xcd:.2f} = busy cycles of one clock domain, from k_spin)", "",
       "| kernel | launches | waves per launch | mean waves in flight | of 8192 slots |", "|---|---|---|---|---|"]
for _, k, nn, w, fl, fr, _v in rows[:24]:
    out.append(f"| `{k[:72]}` | {nn} | {w:.0f} | {fl:.0f} | {100*fr:.1f} % |")
open(sys.argv[2], "w").write("\n".join(out) + "\n")
print("\n".join(out))

#!/usr/bin/env python3
"""Dev tool: where the wave cycles of each kernel go, from one rocprofv3 --pmc pass
(SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD).
All SQ_*_CYCLES / ACTIVE / WAIT counters are quad-cycles summed over waves (MI355X_MICROARCH.md).  Usage: pmc_mix.py <dir>"""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/*/*counter_collection.csv")[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(f)):
    acc[r["Kernel_Name"]][r["Counter_Name"]] += float(r["Counter_Value"])
rows = sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0))
print("| kernel | wave-cycles share | VALU active | LDS active | waiting (s_waitcnt/barrier) | issue stall | VALU / LDS / VMEM-read instructions per wave-kcycle |")
print("|---|---|---|---|---|---|---|")
tot = sum(c.get("SQ_WAVE_CYCLES", 0) for _, c in rows)
for k, c in rows[:18]:
    wc = c.get("SQ_WAVE_CYCLES", 0)
    if wc <= 0: continue
    p = lambda n: 100.0 * c.get(n, 0) / wc
    q = lambda n: 1000.0 * c.get(n, 0) / (4 * wc)
    print(f"| `{k[:60]}` | {100*wc/tot:.1f} % | {p('SQ_ACTIVE_INST_VALU'):.1f} % | {p('SQ_ACTIVE_INST_LDS'):.1f} % | {p('SQ_WAIT_ANY'):.1f} % | {p('SQ_WAIT_INST_ANY'):.1f} % | {q('SQ_INSTS_VALU'):.0f} / {q('SQ_INSTS_LDS'):.0f} / {q('SQ_INSTS_VMEM_RD'):.0f} |")

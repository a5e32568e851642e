#!/usr/bin/env python3
"""Dev tool: GPU busy/idle analysis of a rocprofv3 --kernel-trace CSV (union of kernel intervals)."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows))
t_lo = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5     # analyse the window [t_lo, t_hi] of the trace (fractions)
t_hi = float(sys.argv[3]) if len(sys.argv) > 3 else 0.9
T0, T1 = ev[0][0], max(e[1] for e in ev)
a, b = T0 + (T1 - T0) * t_lo, T0 + (T1 - T0) * t_hi
sel = [e for e in ev if e[0] >= a and e[1] <= b]
busy = 0; cur_s, cur_e = sel[0][0], sel[0][1]; gaps = []
for s, e, n in sel[1:]:
    if s > cur_e:
        busy += cur_e - cur_s; gaps.append((s - cur_e, n)); cur_s, cur_e = s, e
    else: cur_e = max(cur_e, e)
busy += cur_e - cur_s
span = sel[-1][1] - sel[0][0]
print(f"window {span/1e6:.2f} ms, kernels {len(sel)}, busy {busy/1e6:.2f} ms ({100*busy/span:.1f}%), idle {(span-busy)/1e6:.2f} ms")
bykern = collections.Counter()
for g, n in gaps: bykern[n.split("(")[0][:50]] += g
print("idle time before kernel (top):")
for n, g in bykern.most_common(12): print(f"  {g/1e3:9.1f} us  {n}")
big = sorted(gaps, reverse=True)[:8]
print("largest gaps (us):", [(round(g/1e3,1), n.split('(')[0][:30]) for g, n in big])

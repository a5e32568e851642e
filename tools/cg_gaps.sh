#!/bin/bash
# Dev tool (GPU box): the gaps between the five kernels of a V-side CG iteration (end of one -> start of the next), from a
# rocprofv3 kernel trace of the bench workload.  Evidence for DESIGN.md 3.6b ("one persistent kernel per CG iteration").
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/tr && rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -- python3 $GRAFT_REPO_ROOT/bench.py --no-live-traffic --no-cpu --no-f64 --no-profile --steps 10 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
f=$(ls /tmp/tr/*/*kernel_trace.csv | head -1)
python - "$f" <<'PY'
import csv, sys, statistics as st
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0]) for r in rows)
def short(n):
    for k in ("k_sddmm", "k_vsweep_all<float, true", "k_spmm_fin", "k_spmm", "k_cg_bc"):
        if k in n: return k.replace("<float, true", "")
    return None
seq = [(s, e, short(n)) for s, e, n in ev]
gaps, durs = {}, {}
for (s0, e0, a), (s1, e1, b) in zip(seq, seq[1:]):
    if a and b:
        gaps.setdefault(a + " -> " + b, []).append((s1 - e0) / 1e3)
for s, e, a in seq:
    if a: durs.setdefault(a, []).append((e - s) / 1e3)
print("kernel durations (us, median):", {k: round(st.median(v), 1) for k, v in durs.items()})
tot = 0.0
for k, v in sorted(gaps.items(), key=lambda kv: -len(kv[1])):
    if len(v) < 50: continue
    print(f"  {k:32s} n={len(v):4d}  gap median {st.median(v):5.2f} us  mean {st.mean(v):5.2f}  p90 {sorted(v)[int(0.9*len(v))]:5.2f}")
    tot += st.median(v)
print(f"sum of the median gaps around one CG iteration: {tot:.1f} us")
PY

#!/bin/bash
# Dev tool (GPU box): kernel timeline of one outer iteration of the bench workload + busy/idle summary.
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/tr && rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -- python3 $GRAFT_REPO_ROOT/bench.py --no-live-traffic --no-cpu --no-profile --steps 20 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
f=$(ls /tmp/tr/*/*kernel_trace.csv | head -1)
python tools/trace_gaps.py $f 0.5 0.9
python - "$f" <<'EOF'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:48]) for r in rows)
us = [i for i, e in enumerate(ev) if "k_cg_init<" in e[2]]
i0, i1 = us[12], us[13]
t0 = ev[i0][0]; last = t0
for s, e, n in ev[i0:i1]:
    if "k_sddmm" in n or "k_vsweep_all<float, true" in n or "k_spmm" in n or "k_cg_b" in n or "k_cg_c" in n:
        if (s - t0) / 1e3 < 1000: last = max(last, e); continue        # skip the CG body
    print(f"{(s-t0)/1e3:9.1f} {(e-s)/1e3:8.1f}  gap {max(0,(s-last))/1e3:6.1f}  {n}")
    last = max(last, e)
EOF

#!/bin/bash
# Dev tool (GPU box): every kernel of one outer iteration of the bench workload (start, duration, gap to the previous end, queue).
# usage: timeline_full.sh [bench args]
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/tr && rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -- python3 $GRAFT_REPO_ROOT/bench.py --no-live-traffic --no-cpu --no-profile --no-f64 --no-netflix --no-rows --steps 20 "$@" > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
f=$(ls /tmp/tr/*/*kernel_trace.csv | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:56], r.get("Queue_Id", "")) for r in rows)
us = [i for i, e in enumerate(ev) if "k_cg_init<" in e[2]]
i0, i1 = us[12], us[13]
t0 = ev[i0][0]; last = t0
for s, e, n, q in ev[i0:i1]:
    print(f"{(s-t0)/1e3:9.1f} {(e-s)/1e3:8.1f}  gap {max(0,(s-last))/1e3:6.1f}  q{q:>3s}  {n}")
    last = max(last, e)
PY

#!/bin/bash
# the exact-Newton U step (pcr_tune ustep_newton, k_unewton) against the default truncated-CG U step: trajectory and quality on the
# ml1m shape through the CLI, step time through bench.py
mkdir -p gpurun_out
python - <<'PY' 2>&1 | tee gpurun_out/r05_u_newton.log
import re, subprocess, sys, time
sys.path.insert(0, ".")
from primalcr_amd import synth
synth.write_dir(synth.generate("ml1m"), "/tmp/pcr_ml1m")
T = "/root/repo/primalcr_amd/bin/omp-pmf-train"
for name, extra in (("default (truncated CG)", []), ("ustep_newton=1", ["--tune", "ustep_newton=1"]), ("ustep_newton=1 --f64", ["--tune", "ustep_newton=1", "--f64"]), ("default --f64", ["--f64"])):
    p = subprocess.run([T, "-k", "100", "-l", "5000", "-t", "10", "-n", "16", "--timing", *extra, "/tmp/pcr_ml1m", "/tmp/x.model"], cwd="/tmp", capture_output=True, text=True)
    it = re.findall(r"^Iter (\d+) time (\S+) obj (\S+)", p.stdout, re.M)
    te = re.findall(r"^\(Testing\) pairwise error is (\S+) and ndcg is (\S+)", p.stdout, re.M)
    print(f"== {name}: rc {p.returncode}")
    for (i, t, o), (e, n) in zip(it, te):
        print(f"   iter {i:>2s}  time {float(t):8.4f}  obj {o:>12s}  test pairwise error {e}  ndcg@10 {n}")
    print("  ", [l for l in p.stderr.split("\n") if "timing" in l or "rror" in l])
PY
for t in "" "--tune ustep_newton=1"; do python bench.py --no-cpu --no-cli --no-netflix --no-rows --no-live-traffic --no-f64 $t --full-record gpurun_out/r05_u_full.json 2>/dev/null | python -c "import json,sys; l=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('bench ml1m [$t]', l['ms_per_step'], l['ndcg10_test'], l['objective'], l['inner_per_step'], {k: v['wall_us'] for k, v in l['roofline_phase'].items()}, [ (k['slot'], k['avg_us']) for k in l['top_kernels']])"; done 2>&1 | tee -a gpurun_out/r05_u_newton.log

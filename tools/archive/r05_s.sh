#!/bin/bash
# init_s after pcr_initial.cpp moved to g++: CLI --timing on ml1m and Netflix shapes
python - <<'PY'
import subprocess, sys, time
sys.path.insert(0, ".")
from primalcr_amd import synth
synth.write_dir(synth.generate("ml1m"), "/tmp/pcr_ml1m")
synth.write_dir(synth.generate_fast("netflix"), "/tmp/pcr_nf")
T = "/root/repo/primalcr_amd/bin/omp-pmf-train"
for d, n in (("/tmp/pcr_ml1m", 3), ("/tmp/pcr_nf", 2)):
    for rep in range(n):
        t = time.perf_counter()
        p = subprocess.run([T, "-k", "100", "-t", "10", "-n", "16", "--timing", d, "/tmp/x.model"], cwd="/tmp", capture_output=True, text=True)
        print(d, f"wall {time.perf_counter() - t:.3f} s", [l for l in p.stderr.split("\n") if "timing" in l], flush=True)
PY

#!/bin/bash
# A/B: the CLI leaving through _exit after its outputs are flushed, against the plain return (runtime teardown)
python - <<'PY'
import subprocess, sys, time
sys.path.insert(0, ".")
from primalcr_amd import synth
synth.write_dir(synth.generate("ml1m"), "/tmp/pcr_ml1m")
for rep in range(5):
    for exe in ("/root/repo/primalcr_amd/bin/omp-pmf-train", "/root/repo/build_next/omp-pmf-train-plain"):
        t = time.perf_counter()
        p = subprocess.run([exe, "-k", "100", "-t", "10", "-n", "16", "--timing", "/tmp/pcr_ml1m", "/tmp/x.model"], cwd="/tmp", capture_output=True, text=True)
        print(exe.split("/")[-1], f"rc {p.returncode} wall {time.perf_counter() - t:.3f} s", [l.split("wall_s=")[-1] for l in p.stderr.split("\n") if "timing" in l], flush=True)
PY

#!/bin/bash
# the CLI with the HIP runtime brought up on a second thread while the input is parsed: CLI tests, then --timing on ml1m and Netflix shapes
set -o pipefail
mkdir -p gpurun_out
true

python - <<'PY'
import sys
sys.path.insert(0, ".")
from primalcr_amd import synth
synth.write_dir(synth.generate("ml1m"), "/tmp/pcr_ml1m")
synth.write_dir(synth.generate_fast("netflix"), "/tmp/pcr_nf")
PY
python - <<'PY'
import subprocess, time
T = "/root/repo/primalcr_amd/bin/omp-pmf-train"
for d, n in (("/tmp/pcr_ml1m", 4), ("/tmp/pcr_nf", 2)):
    for rep in range(n):
        for extra in ([], ["--no-warmup"]):
            t = time.perf_counter()
            p = subprocess.run([T, "-k", "100", "-t", "10", "-n", "16", "--timing", *extra, d, "/tmp/x.model"], cwd="/tmp", capture_output=True, text=True)
            print(d, extra, f"wall {time.perf_counter() - t:.3f} s", [l for l in p.stderr.split("\n") if "timing" in l], flush=True)
PY

#!/bin/bash
# omp-pmf-predict at size: the CLI tests on the GPU, then the Netflix-shaped test file (4.8 M lines) against a 480 189 x 100 model: ours
# (GPU), ours --host, the reference binary (bounded: its per-line omp region makes it slow)
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests/test_cli.py -m gpu -x -q > gpurun_out/r05_n_clitests.log 2>&1
rc=$?; tail -3 gpurun_out/r05_n_clitests.log; [ $rc -eq 0 ] || exit $rc
python - <<'PY' 2>&1 | tee gpurun_out/r05_n_predict.log
import os, subprocess, sys, time
sys.path.insert(0, ".")
import numpy as np
import primalcr_amd as pcr
from primalcr_amd import synth
R = synth.generate_fast("netflix")
synth.write_dir(R, "/tmp/pcr_nf")
pcr.model_save("/tmp/nf.model", pcr.initial(R.d1, 100) * 0.1, pcr.initial(R.d2, 100) * 0.1)
test = "/tmp/pcr_nf/test.ratings"
print("test file:", os.path.getsize(test) / 1e6, "MB,", len(R.tval), "lines", flush=True)
runs = [("ours (GPU)", [os.path.abspath("primalcr_amd/bin/omp-pmf-predict")], "/tmp/o1.txt"),
        ("ours --host", [os.path.abspath("primalcr_amd/bin/omp-pmf-predict"), "--host"], "/tmp/o2.txt"),
        ("reference -- OMP_NUM_THREADS=1", [os.path.abspath("oracle/_ref/omp-pmf-predict")], "/tmp/o3.txt")]
for name, cmd, out in runs:
    t = time.perf_counter()
    env = dict(os.environ, OMP_NUM_THREADS="1") if "reference" in name else None
    try:
        p = subprocess.run(cmd + [test, "/tmp/nf.model", out], capture_output=True, text=True, timeout=600, env=env)
        print(f"{name}: rc {p.returncode}, {time.perf_counter() - t:.2f} s wall {p.stderr[-200:]}", flush=True)
    except subprocess.TimeoutExpired:
        print(f"{name}: more than 600 s", flush=True)
a, b = open("/tmp/o2.txt").read(), open("/tmp/o3.txt").read() if os.path.exists("/tmp/o3.txt") else None
print("--host == reference bytes:", a == b)
g = np.loadtxt("/tmp/o1.txt"); h = np.loadtxt("/tmp/o2.txt")
print("GPU vs host: max abs diff", float(np.abs(g - h).max()), "lines", len(g))
PY

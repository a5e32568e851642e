#!/bin/bash
# round 5, second GPU call: six ranks on the one GPU (the case that died silently in round 4) through both entry points, then the
# Netflix-shaped CLI run end to end
set -o pipefail
mkdir -p gpurun_out
python bench.py --gpus 6 --comm p2p --devices 0,0,0,0,0,0 --rendezvous gloo --users 500 --nnz 60000 --steps 2 --warmup 1 --no-cpu --no-f64 --no-rows \
    --full-record gpurun_out/r05_b_6rank_full.json > gpurun_out/r05_b_6rank_line.json 2> gpurun_out/r05_b_6rank.err
echo "bench --gpus 6: rc $?"; tail -3 gpurun_out/r05_b_6rank.err; head -c 600 gpurun_out/r05_b_6rank_line.json; echo
python - <<'PY' > gpurun_out/r05_b_cli6.log 2>&1
import os, subprocess, sys, numpy as np
sys.path.insert(0, ".")
from primalcr_amd import synth
R = synth.generate("ml1m", d1=1200, nnz=150000)
d = synth.write_dir(R, "/tmp/pcr_cli6/data")
T = os.path.abspath("primalcr_amd/bin/omp-pmf-train")
base = [T, "-k", "16", "-l", "500", "-t", "3", "--f64", "-p", "1"]
one = subprocess.run(base + [d, "one.model"], cwd="/tmp/pcr_cli6", capture_output=True, text=True)
six = subprocess.run(base + ["--gpus", "6", "--devices", "0,0,0,0,0,0", "--comm", "p2p", d, "six.model"], cwd="/tmp/pcr_cli6", capture_output=True, text=True)
print("one rc", one.returncode, "six rc", six.returncode)
print(six.stderr[-1500:])
f = lambda t: [l for l in t.split("\n") if l.startswith(("Iter", "(T"))]
a, b = f(one.stdout), f(six.stdout)
print("\n".join(b))
print("printed lines equal:", [x.split(" time ")[0] + x.split(" obj")[-1] if x.startswith("Iter") else x for x in a] == [x.split(" time ")[0] + x.split(" obj")[-1] if x.startswith("Iter") else x for x in b])
ma, mb = np.fromfile("/tmp/pcr_cli6/one.model"), np.fromfile("/tmp/pcr_cli6/six.model")
print("model max rel diff:", float(np.nanmax(np.abs(ma[2:] - mb[2:])) / np.nanmax(np.abs(ma[2:]))))
PY
echo "cli --gpus 6: rc $?"; tail -12 gpurun_out/r05_b_cli6.log
python tools/exp_cli_e2e.py --shape netflix --out gpurun_out/r05_cli_netflix.json > gpurun_out/r05_b_cli_netflix.log 2>&1
echo "netflix cli: rc $?"; tail -50 gpurun_out/r05_b_cli_netflix.log

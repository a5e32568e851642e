#!/bin/bash
# (1) set-up phases of pcr_solver_create on the ml1m shape; (2) the drop-in CLI on the WHOLE Yahoo!Music-shaped text directory
# (700 M ratings, ~11 GB of text), one iteration without evaluation: load / init / create / write at size
mkdir -p gpurun_out
python - <<'PY'
import sys, time
sys.path.insert(0, ".")
from primalcr_amd import synth
R = synth.generate("ml1m")
synth.write_dir(R, "/tmp/pcr_ml1m")
PY
(cd /tmp && /root/repo/primalcr_amd/bin/omp-pmf-train -k 100 -t 10 -n 16 --timing --tune debug=1 /tmp/pcr_ml1m /tmp/ml1m.model > /dev/null 2> /root/repo/gpurun_out/r05_j_ml1m.err; grep "set-up\|timing" /root/repo/gpurun_out/r05_j_ml1m.err)
df -h /tmp | tail -1
python - <<'PY' 2>&1 | tee gpurun_out/r05_j_yahoo_cli.log
import json, os, subprocess, sys, time
sys.path.insert(0, ".")
from primalcr_amd import synth
t = time.time()
R = synth.generate_fast("yahoo")
t1 = time.time()
print(f"generated {R.nnz} ratings in {t1 - t:.1f}s", flush=True)
synth.write_dir(R, "/tmp/pcr_yahoo")
size = sum(os.path.getsize(os.path.join("/tmp/pcr_yahoo", f)) for f in os.listdir("/tmp/pcr_yahoo"))
print(f"text directory: {size / 1e9:.2f} GB written in {time.time() - t1:.1f}s", flush=True)
del R
t0 = time.time()
p = subprocess.run([os.path.abspath("primalcr_amd/bin/omp-pmf-train"), "-k", "200", "-t", "1", "-p", "0", "-n", "16", "--timing", "--tune", "debug=1", "/tmp/pcr_yahoo", "/tmp/yahoo.model"],
                   cwd="/tmp", capture_output=True, text=True, timeout=900)
print("rc", p.returncode, f"wall {time.time() - t0:.1f}s")
print(p.stdout[-600:])
print("\n".join(l for l in p.stderr.split("\n") if "set-up" in l or "timing" in l or "rror" in l))
PY
rm -rf /tmp/pcr_yahoo /tmp/yahoo.model /tmp/U.txt /tmp/V.txt

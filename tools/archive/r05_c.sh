#!/bin/bash
# set-up phases of pcr_solver_create on the Netflix shape (pcr_tune debug=1) through the CLI
mkdir -p gpurun_out
python - <<'PY'
import sys, time
sys.path.insert(0, ".")
from primalcr_amd import synth
t = time.time()
R = synth.generate_fast("netflix")
synth.write_dir(R, "/tmp/pcr_nf")
print("data", time.time() - t, file=sys.stderr)
PY
cd /tmp && /root/repo/primalcr_amd/bin/omp-pmf-train -k 100 -t 1 -p 1 -n 16 --timing --tune debug=1 /tmp/pcr_nf /tmp/nf.model > /root/repo/gpurun_out/r05_c_create.out 2> /root/repo/gpurun_out/r05_c_create.err
echo rc $?; grep "set-up\|timing\|lanes" /root/repo/gpurun_out/r05_c_create.err

#!/bin/bash
# where pcr_solver_create's time goes on the ml1m shape (pcr_tune debug=1), three processes in a row
python - <<'PY'
import sys
sys.path.insert(0, ".")
from primalcr_amd import synth
synth.write_dir(synth.generate("ml1m"), "/tmp/pcr_ml1m")
PY
cd /tmp
for i in 1 2 3; do /root/repo/primalcr_amd/bin/omp-pmf-train -k 100 -t 10 -n 16 --timing --tune debug=1 /tmp/pcr_ml1m /tmp/ml1m.model 2>&1 >/dev/null | grep "set-up\|timing"; echo; done

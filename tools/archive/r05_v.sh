#!/bin/bash
# counter evidence for k_unewton (the optional exact-Newton U step): fp64 MFMA instructions per launch, in a --pmc pass of its own
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
cd /tmp && rocprofv3 --pmc SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $R/gpurun_out/r05_v_pmc -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu --no-cli --no-f64 --no-netflix --no-rows --no-profile --no-live-traffic --tune ustep_newton=1 > /dev/null 2> $R/gpurun_out/r05_v.err || { tail -5 $R/gpurun_out/r05_v.err; exit 1; }
python3 $R/tools/pmc_table.py $R/gpurun_out/r05_v_pmc > $R/gpurun_out/r05_v_pmc.json
rm -rf $R/gpurun_out/r05_v_pmc
python3 - <<PY
import json
t = json.load(open("$R/gpurun_out/r05_v_pmc.json"))
for k, v in t.items():
    if "unewton" in k or "ustep_gram" in k:
        print(k[:60], {a: b for a, b in v.items()})
PY

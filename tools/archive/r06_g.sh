#!/bin/bash
# round 6, seventh GPU call: an upper bound of what a faster cluster class could give the U step (item 4); k_unewton's matrix-core
# and L2 counters (item 5); the N = 2 line with the full-size Netflix-shaped leg on one GPU; the CLI's set-up split, three runs
set -o pipefail
mkdir -p gpurun_out/r06_g
export TMPDIR=/tmp
R=$PWD
python tools/exp_ustep_without_long.py > gpurun_out/r06_g/ustep_without_long.txt 2>&1 || { tail -5 gpurun_out/r06_g/ustep_without_long.txt; exit 1; }
cut -c1-330 gpurun_out/r06_g/ustep_without_long.txt
B="$R/bench.py --steps 4 --warmup 2 --no-cpu --no-cli --no-f64 --no-netflix --no-rows --no-profile --no-live-traffic --no-hbm --tune ustep_newton=1"
cd /tmp
for C in "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_LDS" "GRBM_GUI_ACTIVE" "TCP_TCC_READ_REQ_sum TCC_REQ_sum" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_DRAM_32B_sum TCC_EA0_WRREQ_WRITE_DRAM_32B_sum"; do
    tag=$(echo $C | cut -d' ' -f1)
    rocprofv3 --pmc $C --output-format csv -d $R/gpurun_out/r06_g/pmc_$tag -- python3 $B > /dev/null 2> $R/gpurun_out/r06_g/pmc_$tag.err || { tail -3 $R/gpurun_out/r06_g/pmc_$tag.err; exit 1; }
    python3 $R/tools/pmc_table.py $R/gpurun_out/r06_g/pmc_$tag > $R/gpurun_out/r06_g/pmc_$tag.json; rm -rf $R/gpurun_out/r06_g/pmc_$tag
done
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r06_g/trace -- python3 $B > /dev/null 2> $R/gpurun_out/r06_g/trace.err; cp $(ls $R/gpurun_out/r06_g/trace/*/*kernel_stats.csv | head -1) $R/gpurun_out/r06_g/newton_kernel_stats.csv; rm -rf $R/gpurun_out/r06_g/trace
cd $R
python3 - <<PY
import json, glob
for f in sorted(glob.glob("gpurun_out/r06_g/pmc_*.json")):
    t = json.load(open(f))
    for k, v in t.items():
        if "unewton" in k: print(f.split("pmc_")[1], {a: b for a, b in v.items()})
PY
grep unewton gpurun_out/r06_g/newton_kernel_stats.csv | cut -c1-200
python bench.py --tune ustep_newton=1 --no-cpu --no-cli --no-f64 --no-netflix --no-rows --no-live-traffic --no-hbm --full-record gpurun_out/r06_g/newton_bench_full.json > gpurun_out/r06_g/newton_line.json 2> gpurun_out/r06_g/newton_bench.err
python -c "
import json; l=json.load(open('gpurun_out/r06_g/newton_line.json')); print('newton bench', l['ms_per_step'], l['roofline_phase'])"
python bench.py --gpus 2 --devices 0,0 --comm p2p --rendezvous gloo --no-cpu --full-record gpurun_out/r06_g/two_ranks_full.json > gpurun_out/r06_g/two_ranks_line.json 2> gpurun_out/r06_g/two_ranks.err || { tail -20 gpurun_out/r06_g/two_ranks.err; exit 1; }
wc -c gpurun_out/r06_g/two_ranks_line.json; python -c "
import json; l=json.load(open('gpurun_out/r06_g/two_ranks_line.json')); print(l['value'], l['ms_per_step'], l.get('exchange')); print(l['netflix'])"
d=$(mktemp -d /tmp/cli_XXXX); python -c "
import sys; sys.path.insert(0, '$R')
from primalcr_amd import synth
synth.write_dir(synth.generate('ml1m', seed=synth.SEED), '$d/data')"
for i in 1 2 3; do (cd $d && $R/primalcr_amd/bin/omp-pmf-train -k 100 -l 5000 -t 10 -n 16 --timing data m.model 2>&1 | grep timing); done | tee gpurun_out/r06_g/cli_timing.txt

#!/bin/bash
# round 6: k_unewton with the blocked Cholesky -- its parity tests, then the step time and the kernel's average under rocprofv3
set -o pipefail
mkdir -p gpurun_out/r06_i
export TMPDIR=/tmp
R=$PWD
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "newton" > gpurun_out/r06_i_tests.$(date +%s).log 2>&1
rc=$?; tail -15 gpurun_out/r06_i_tests.*.log | cut -c1-300; [ $rc -eq 0 ] || exit $rc
python bench.py --tune ustep_newton=1 --no-cpu --no-cli --no-f64 --no-netflix --no-rows --no-live-traffic --no-hbm --full-record gpurun_out/r06_i/newton_bench_full.json > gpurun_out/r06_i/newton_line.json 2> gpurun_out/r06_i/newton_bench.err || { tail -5 gpurun_out/r06_i/newton_bench.err; exit 1; }
python -c "
import json; l=json.load(open('gpurun_out/r06_i/newton_line.json')); print('newton bench', l['ms_per_step'], l['ndcg10_test'], l['objective'], l['roofline_phase'])"
cd /tmp
B="$R/bench.py --steps 4 --warmup 2 --no-cpu --no-cli --no-f64 --no-netflix --no-rows --no-profile --no-live-traffic --no-hbm --tune ustep_newton=1"
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r06_i/trace -- python3 $B > /dev/null 2> $R/gpurun_out/r06_i/trace.err; grep unewton $(ls $R/gpurun_out/r06_i/trace/*/*kernel_stats.csv | head -1) | cut -c1-230; rm -rf $R/gpurun_out/r06_i/trace
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d $R/gpurun_out/r06_i/pmc -- python3 $B > /dev/null 2> $R/gpurun_out/r06_i/pmc.err && python3 $R/tools/pmc_table.py $R/gpurun_out/r06_i/pmc | python3 -c "
import json,sys
t=json.load(sys.stdin)
for k,v in t.items():
    if 'unewton' in k: print(v)"
rm -rf $R/gpurun_out/r06_i/pmc
cd $R; python tools/exp_newton_prof.py 2>&1 | grep k_unewton | tail -4

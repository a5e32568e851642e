#!/bin/bash
# round 6, third GPU call: the queue probe with queues created while peers spin; the SMU's accumulating activity counters on known
# patterns; the multi-rank tests with the queue-budget guard; the default bench with the HBM side; the Yahoo-shaped share
set -o pipefail
REPO=$PWD
mkdir -p gpurun_out/r06_c
export TMPDIR=/tmp
timeout -k 10 200 tools/ubench/_build/queue_budget_probe 5 4,6,9 5 late > gpurun_out/r06_c/queue_budget_late.txt 2>&1; echo "probe rc $?"
cat gpurun_out/r06_c/queue_budget_late.txt
for P in idle read reread gather109 gather1024 write; do
    python3 tools/umc_sample.py --out gpurun_out/r06_c/umc_calib_acc.jsonl -- tools/ubench/_build/dram_calib loop $P 2 > /dev/null || exit 1
done
python3 -c "
import json
for l in open('gpurun_out/r06_c/umc_calib_acc.jsonl'):
    r = json.loads(l); c = list(r['cards'].values())[0]
    print(r['stdout_json']['pattern'], r['stdout_json']['known_GBs'], c['mem_busy_mean'], r.get('accumulators'))
"
python -m pytest tests/test_cli.py tests/test_bench_entry.py tests/test_gpu_system.py -x -q -m gpu -k "gpus_option or ranks or netflix or p2p or rccl" > gpurun_out/r06_c_tests.$(date +%s).log 2>&1
rc=$?; tail -5 gpurun_out/r06_c_tests.*.log; [ $rc -eq 0 ] || exit $rc
python bench.py --full-record gpurun_out/r06_c_bench_full.json > gpurun_out/r06_c_line.json 2> gpurun_out/r06_c_bench.err || { tail -20 gpurun_out/r06_c_bench.err; exit 1; }
grep -E "traffic|hbm|cli\]" gpurun_out/r06_c_bench.err | tail -8; wc -c gpurun_out/r06_c_line.json
python bench.py --shape yahoo --steps 3 --warmup 1 --no-cpu --no-f64 --full-record gpurun_out/r06_c_yahoo_share_full.json > gpurun_out/r06_c_yahoo_share_line.json 2> gpurun_out/r06_c_yahoo_share.err || { tail -20 gpurun_out/r06_c_yahoo_share.err; exit 1; }
python -c "
import json
for f in ('gpurun_out/r06_c_line.json', 'gpurun_out/r06_c_yahoo_share_line.json'):
    l = json.load(open(f)); print(f, l['ms_per_step'], l['roofline'], l.get('hbm'))
    for k in ('f64', 'netflix'):
        if l.get(k): print('  ', k, l[k].get('ms_per_step'), l[k].get('hbm'))
"

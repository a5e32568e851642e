#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "blocked_user or launch_variants" > gpurun_out/r05_w_vblock.log 2>&1
rc=$?; tail -30 gpurun_out/r05_w_vblock.log | cut -c1-250; [ $rc -eq 0 ] || exit $rc
for t in "" "--tune vblock_users=64" "--tune vblock_users=256"; do python bench.py --no-cpu --no-cli --no-netflix --no-rows --no-live-traffic --no-f64 $t 2>/dev/null | python -c "import json,sys; l=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('bench ml1m [$t]', l['ms_per_step'], l['ndcg10_test'], l['objective'], {k: v['wall_us'] for k, v in l['roofline_phase'].items()}, [(k['slot'], k['avg_us']) for k in l['top_kernels']])"; done

#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "exact_newton" > gpurun_out/r05_t_newton.log 2>&1
rc=$?; tail -30 gpurun_out/r05_t_newton.log | cut -c1-250; [ $rc -eq 0 ] || exit $rc
python bench.py --no-cpu --no-cli --no-netflix --no-rows --no-live-traffic --no-f64 --full-record gpurun_out/r05_t_full.json 2>/dev/null | python -c "import json,sys; l=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('ml1m', l['ms_per_step'], l['roofline_phase'])"

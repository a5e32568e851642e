#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r05_q_gputests.log 2>&1
rc=$?; tail -4 gpurun_out/r05_q_gputests.log; [ $rc -eq 0 ] || exit $rc
for i in 1 2; do python bench.py --no-cpu --no-cli --no-netflix --no-rows --no-live-traffic --full-record gpurun_out/r05_q_full_$i.json 2>/dev/null | python -c "import json,sys; l=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('ml1m', l['ms_per_step'], l['f64']['ms_per_step'], l['solver_create_s'], l['roofline_phase'])"; done
bash tools/r05_p.sh | tail -14

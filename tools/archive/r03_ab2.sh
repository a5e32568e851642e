#!/bin/bash
# Dev tool (GPU box): A/B of launch knobs with the driver's command line (--steps 20 --warmup 5, event timing on).
# usage: r03_ab2.sh "<tune args 1>" "<tune args 2>" ...
for cfg in "$@"; do
  for rep in 1 2 3; do
    python3 bench.py --no-live-traffic --full-line --gpus 1 --steps 20 --warmup 5 --no-cpu --no-f64 --no-netflix --no-rows $cfg 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('%-60s %.4f ms  u-step wall %.1f us' % ('$cfg', d['ms_per_step'], d['roofline_phase']['u_step']['wall_us_per_step']))"
  done
done

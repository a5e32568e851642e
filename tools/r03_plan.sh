#!/bin/bash
# Dev tool (GPU box): A/B of U-step placements (pcr_tune "uplan") on the ml1m bench workload, speculation off.
# usage: r03_plan.sh "<uplan 1>" "<uplan 2>" ...   ("" = the built-in plan)
for cfg in "$@"; do
  t=""; [ -n "$cfg" ] && t="--tune uplan=$cfg"
  for rep in 1 2; do
    python3 bench.py --no-cpu --no-f64 --no-netflix --no-rows --no-profile --steps 40 --warmup 5 --tune speculate=0 $t 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('%-50s %.4f ms' % ('$cfg', d['ms_per_step']))"
  done
done

#!/bin/bash
# Dev tool (GPU box): A/B of launch knobs on the ml1m bench workload.  usage: r03_ab.sh "<tune args 1>" "<tune args 2>" ...
# each argument is a (possibly empty) list of --tune key=value options; prints ms per step without event timing, 3 repeats each
for cfg in "$@"; do
  for rep in 1 2 3; do
    python3 bench.py --no-live-traffic --no-cpu --no-f64 --no-netflix --no-rows --no-profile --steps 40 --warmup 5 $cfg 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('%-60s %.4f ms' % ('$cfg', d['ms_per_step']))"
  done
done

#!/bin/bash
# Dev tool (GPU box): A/B of launch knobs on a large shape.  usage: r03_ab_shape.sh <shape> <steps> "<tune args 1>" "<tune args 2>" ...
SHAPE=$1; STEPS=$2; shift 2
for cfg in "$@"; do
  python3 bench.py --no-live-traffic --full-line --shape $SHAPE --no-cpu --no-f64 --no-rows --steps $STEPS --warmup 2 $cfg 2>/dev/null | python3 -c "
import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); ph=d['roofline_phase']
pk={k:v['avg_us'] for k,v in d['kernels'].items() if k.startswith('prepare')}
print('%-44s %9.3f ms/step  u_step %9.1f us  v_step %9.1f us  prepare %s' % ('$cfg', d['ms_per_step'], ph['u_step']['wall_us_per_step'], ph['v_step']['wall_us_per_step'], pk))"
done

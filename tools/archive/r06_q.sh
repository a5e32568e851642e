#!/bin/bash
# round 6: fp64, ratings per lane group of k_spmm (spmm_chunk) -- the one knob of the fp64 sweep that showed outside the scatter
set -o pipefail
mkdir -p gpurun_out/r06_q
B="python bench.py --precision f64 --steps 20 --warmup 5 --no-cpu --no-cli --no-netflix --no-rows --no-live-traffic --no-hbm"
run() { tag=$1; shift; $B "$@" --full-record gpurun_out/r06_q/$tag.json > gpurun_out/r06_q/$tag.line 2> gpurun_out/r06_q/$tag.err || { echo "$tag failed"; return; }
  python -c "
import json
l=json.load(open('gpurun_out/r06_q/$tag.line')); ph=l.get('roofline_phase') or {}; f=json.load(open('gpurun_out/r06_q/$tag.json'))
print('%-14s %.4f ms (no events %.4f)  U %s  V %s  spmm %s' % ('$tag', l['ms_per_step'], f.get('ms_per_step_noevents') or 0, (ph.get('u_step') or {}).get('wall_us'), (ph.get('v_step') or {}).get('wall_us'), f['kernels']['spmm']['avg_us']))"; }
for rep in 1 2; do
  run default_$rep
  for c in 48 64 80 96; do run chunk${c}_$rep --tune spmm_chunk=$c; done
done

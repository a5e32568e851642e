#!/bin/bash
# round 6: fp32 ml1m, the SpMM's chunk at a finer rounding (80 / 88 fill the one round better than 96)
set -o pipefail
mkdir -p gpurun_out/r06_s
B="python bench.py --steps 20 --warmup 5 --no-cpu --no-cli --no-netflix --no-rows --no-live-traffic --no-hbm --no-f64"
run() { tag=$1; shift; $B "$@" --full-record gpurun_out/r06_s/$tag.json > gpurun_out/r06_s/$tag.line 2> gpurun_out/r06_s/$tag.err || { echo "$tag failed"; return; }
  python -c "
import json
l=json.load(open('gpurun_out/r06_s/$tag.line')); f=json.load(open('gpurun_out/r06_s/$tag.json'))
print('%-14s %.4f ms (no events %.4f)  spmm %s us  fin %s us' % ('$tag', l['ms_per_step'], f.get('ms_per_step_noevents') or 0, f['kernels']['spmm']['avg_us'], f['kernels']['spmm_fin']['avg_us']))"; }
for rep in 1 2 3; do
  run default_$rep
  run chunk80_$rep --tune spmm_chunk=80
  run chunk88_$rep --tune spmm_chunk=88
done

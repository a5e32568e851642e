#!/bin/bash
# round 6: the SpMM's chunk by the fullest last round (plan_pick_chunk) -- the tests that walk the plan, then fp64 / fp32 / a 2 M-rating shard
set -o pipefail
mkdir -p gpurun_out/r06_r
python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py -x -q -m gpu -k "variants or v_side or fuzz or sharding or netflix or scale or golden" > gpurun_out/r06_r_tests.$(date +%s).log 2>&1
rc=$?; tail -3 gpurun_out/r06_r_tests.*.log | cut -c1-200; [ $rc -eq 0 ] || exit $rc
B="python bench.py --steps 20 --warmup 5 --no-cpu --no-cli --no-netflix --no-rows --no-live-traffic --no-hbm"
run() { tag=$1; shift; $B "$@" --full-record gpurun_out/r06_r/$tag.json > gpurun_out/r06_r/$tag.line 2> gpurun_out/r06_r/$tag.err || { echo "$tag failed"; tail -3 gpurun_out/r06_r/$tag.err; return; }
  python -c "
import json
l=json.load(open('gpurun_out/r06_r/$tag.line')); f=json.load(open('gpurun_out/r06_r/$tag.json'))
print('%-22s %.4f ms (no events %.4f)  spmm %s us' % ('$tag', l['ms_per_step'], f.get('ms_per_step_noevents') or 0, f['kernels']['spmm']['avg_us']))"; }
for rep in 1 2; do
  run f64_new_$rep --precision f64 --no-f64
  run f64_128_$rep --precision f64 --no-f64 --tune spmm_chunk=128
  run f32_new_$rep --no-f64
  run f32_2M_new_$rep --no-f64 --users 12000 --nnz 2000000
  run f32_2M_128_$rep --no-f64 --users 12000 --nnz 2000000 --tune spmm_chunk=128
done

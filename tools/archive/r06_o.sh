#!/bin/bash
# round 6: less-travelled flag combinations of bench.py once each (stand-alone Netflix shape, fp64 as the main precision, no profile, toy sizes)
set -o pipefail
mkdir -p gpurun_out/r06_o
run() { tag=$1; shift; python bench.py "$@" --full-record gpurun_out/r06_o/$tag.json > gpurun_out/r06_o/$tag.line 2> gpurun_out/r06_o/$tag.err || { echo "$tag FAILED"; tail -5 gpurun_out/r06_o/$tag.err | cut -c1-300; return; }
  python3 -c "
import json; l=json.load(open('gpurun_out/r06_o/$tag.line')); print('%-18s %.4f ms  hbm %s  traffic %s  bytes %d' % ('$tag', l['ms_per_step'], (l.get('hbm') or {}).get('frac'), (l.get('roofline') or {}).get('traffic_source'), len(open('gpurun_out/r06_o/$tag.line').read())))"; }
run netflix_alone --shape netflix --steps 3 --warmup 1 --no-cpu --no-f64
run f64_main --precision f64 --no-cpu --no-cli --no-netflix
run no_profile --no-profile --no-cpu --no-cli --no-netflix --no-f64
run small --users 500 --nnz 40000 --steps 3 --warmup 1 --no-cpu
run no_hbm --no-hbm --no-live-traffic --no-cpu --no-cli --no-netflix --no-f64

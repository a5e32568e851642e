#!/bin/bash
# round 6: configs[4] itself -- the whole Yahoo!Music shape (1.8 M x 136 k, 700 M ratings, k = 200) -- on ONE GPU with the HBM side and the
# live traffic pass; a progress line every minute (the generation and the counter pass are minutes each)
set -o pipefail
mkdir -p gpurun_out/r06_l
( while true; do sleep 55; echo "[r06_l] $(date +%T) still running"; done ) &
HB=$!
python bench.py --shape yahoo --users 1800000 --steps 2 --warmup 1 --no-cpu --no-f64 --no-rows --full-record gpurun_out/r06_l/yahoo_full.json > gpurun_out/r06_l/yahoo_line.json 2> gpurun_out/r06_l/yahoo.err
rc=$?
kill $HB
tail -6 gpurun_out/r06_l/yahoo.err | cut -c1-300
[ $rc -eq 0 ] || exit $rc
python -c "
import json; l=json.load(open('gpurun_out/r06_l/yahoo_line.json')); print(l['ms_per_step'], l['solver_create_s'], l['ndcg10_test'], l.get('hbm'), l['roofline'])"

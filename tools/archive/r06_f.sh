#!/bin/bash
# round 6, sixth GPU call: the two-stream form of a Hessian-vector product, timing only (probe build); the default bench with the
# HBM replay that repeats the measurement's own iterations
set -o pipefail
mkdir -p gpurun_out/r06_f
for P in f32 f64; do python tools/exp_pipe_probe.py --precision $P 2>&1 | grep pipe-probe | tee -a gpurun_out/r06_f/pipe_probe.txt; done
python tools/exp_pipe_probe.py --shape netflix 2>&1 | grep pipe-probe | tee -a gpurun_out/r06_f/pipe_probe.txt
python bench.py --full-record gpurun_out/r06_f_bench_full.json > gpurun_out/r06_f_line.json 2> gpurun_out/r06_f_bench.err || { tail -20 gpurun_out/r06_f_bench.err; exit 1; }
python bench.py --shape yahoo --steps 3 --warmup 1 --no-cpu --no-f64 --full-record gpurun_out/r06_f_yahoo_share_full.json > gpurun_out/r06_f_yahoo_share_line.json 2> gpurun_out/r06_f_yahoo_share.err || { tail -20 gpurun_out/r06_f_yahoo_share.err; exit 1; }
python -c "
import json
for f in ('gpurun_out/r06_f_line.json', 'gpurun_out/r06_f_yahoo_share_line.json'):
    l = json.load(open(f)); print(f, l['ms_per_step'], l.get('hbm'))
    for k in ('f64', 'netflix'):
        if l.get(k): print('  ', k, l[k].get('ms_per_step'), l[k].get('hbm'))
full = json.load(open('gpurun_out/r06_f_bench_full.json')); print(full['hbm']['replay_ms_per_step'], full['f64']['hbm']['replay_ms_per_step'], full['netflix']['hbm']['replay_ms_per_step'])
"

#!/bin/bash
# round 6, fifth GPU call: fork / join cost between two streams; the default bench and the Yahoo-shaped share with the unperturbed
# HBM replay; the round's profile set (kernel stats, exact traffic counters, L2 hit rate, occupancy, wave-cycle mix)
set -o pipefail
mkdir -p gpurun_out/r06_e
export TMPDIR=/tmp
tools/ubench/_build/join_probe > gpurun_out/r06_e/join_probe.txt 2>&1; cat gpurun_out/r06_e/join_probe.txt
python bench.py --full-record gpurun_out/r06_e_bench_full.json > gpurun_out/r06_e_line.json 2> gpurun_out/r06_e_bench.err || { tail -20 gpurun_out/r06_e_bench.err; exit 1; }
python bench.py --shape yahoo --steps 3 --warmup 1 --no-cpu --no-f64 --full-record gpurun_out/r06_e_yahoo_share_full.json > gpurun_out/r06_e_yahoo_share_line.json 2> gpurun_out/r06_e_yahoo_share.err || { tail -20 gpurun_out/r06_e_yahoo_share.err; exit 1; }
python -c "
import json
for f in ('gpurun_out/r06_e_line.json', 'gpurun_out/r06_e_yahoo_share_line.json'):
    l = json.load(open(f)); print(f, l['ms_per_step'], l.get('hbm'))
    for k in ('f64', 'netflix'):
        if l.get(k): print('  ', k, l[k].get('ms_per_step'), l[k].get('hbm'))
"
bash tools/collect_profiles.sh r06_e ml1m f32 "trace dram l2 lds occ mix" > gpurun_out/r06_e/collect_f32.log 2>&1 || { tail -5 gpurun_out/r06_e/collect_f32.log; exit 1; }
bash tools/collect_profiles.sh r06_e ml1m f64 "trace dram l2 occ" > gpurun_out/r06_e/collect_f64.log 2>&1 || { tail -5 gpurun_out/r06_e/collect_f64.log; exit 1; }
bash tools/collect_profiles.sh r06_e netflix f32 "trace dram l2 occ" --steps 3 --warmup 1 > gpurun_out/r06_e/collect_nf.log 2>&1 || { tail -5 gpurun_out/r06_e/collect_nf.log; exit 1; }
bash tools/collect_profiles.sh r06_e yahoo f32 "trace dram l2" --steps 2 --warmup 1 > gpurun_out/r06_e/collect_yh.log 2>&1 || { tail -5 gpurun_out/r06_e/collect_yh.log; exit 1; }
ls gpurun_out/r06_e/*

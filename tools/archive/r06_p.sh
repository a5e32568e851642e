#!/bin/bash
# round 6, last call: smoke(), the whole -m gpu suite and the driver's bench command on the tree as committed
set -o pipefail
mkdir -p gpurun_out/r06_p
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06_p/smoke.txt 2>&1 || { tail -5 gpurun_out/r06_p/smoke.txt; exit 1; }
tail -1 gpurun_out/r06_p/smoke.txt
python -m pytest tests -x -q -m gpu > gpurun_out/r06_p_gputests.$(date +%s).log 2>&1
rc=$?; tail -4 gpurun_out/r06_p_gputests.*.log | cut -c1-300; [ $rc -eq 0 ] || exit $rc
python bench.py --full-record gpurun_out/r06_p_bench_full.json > gpurun_out/r06_p_line.json 2> gpurun_out/r06_p_bench.err || { tail -20 gpurun_out/r06_p_bench.err; exit 1; }
wc -c gpurun_out/r06_p_line.json; python -c "
import json; l=json.load(open('gpurun_out/r06_p_line.json')); print(l['ms_per_step'], l['value'], l['roofline']['frac'], l['hbm'], l['f64']['ms_per_step'], l['netflix']['ms_per_step'], l['cli']['wall_s'], l['cli']['create_s'])"

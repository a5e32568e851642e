#!/bin/bash
# end-of-round verification: the -m gpu suite, the driver's bench command, the Yahoo-shaped share and the Netflix shape as stand-alone records
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r05_o_gputests.log 2>&1
rc=$?; tail -4 gpurun_out/r05_o_gputests.log; [ $rc -eq 0 ] || exit $rc
python bench.py --full-record gpurun_out/r05_o_bench_full.json > gpurun_out/r05_o_line.json 2> gpurun_out/r05_o_bench.err || { tail -20 gpurun_out/r05_o_bench.err; exit 1; }
tail -3 gpurun_out/r05_o_bench.err; wc -c gpurun_out/r05_o_line.json
python bench.py --shape yahoo --steps 3 --warmup 1 --no-cpu --no-f64 --full-record gpurun_out/r05_o_yahoo_share_full.json > gpurun_out/r05_o_yahoo_share_line.json 2> gpurun_out/r05_o_yahoo_share.err || { tail -20 gpurun_out/r05_o_yahoo_share.err; exit 1; }
python -c "import json; l=json.load(open('gpurun_out/r05_o_yahoo_share_line.json')); print('yahoo share', l['ms_per_step'], l['solver_create_s'], l['gather'])"

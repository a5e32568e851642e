#!/bin/bash
# end of round 5: the driver's bench command; the Netflix-shaped CLI end to end (ours only: the reference's figure is in
# profiles/r05_cli_netflix.json); the whole Yahoo!Music shape on one GPU (solver_create_s at 700 M ratings)
set -o pipefail
mkdir -p gpurun_out
python bench.py --full-record gpurun_out/r05_i_bench_full.json > gpurun_out/r05_i_line.json 2> gpurun_out/r05_i_bench.err || { tail -20 gpurun_out/r05_i_bench.err; exit 1; }
tail -4 gpurun_out/r05_i_bench.err; wc -c gpurun_out/r05_i_line.json
python tools/exp_cli_e2e.py --shape netflix --no-reference --out gpurun_out/r05_cli_netflix_end.json > gpurun_out/r05_i_cli_netflix.log 2>&1 || { tail -20 gpurun_out/r05_i_cli_netflix.log; exit 1; }
grep '"wall_s"\|_s"' gpurun_out/r05_cli_netflix_end.json
python bench.py --shape yahoo --users 1800000 --steps 2 --warmup 1 --no-cpu --no-f64 --no-rows --full-record gpurun_out/r05_i_yahoo_full.json > gpurun_out/r05_i_yahoo_line.json 2> gpurun_out/r05_i_yahoo.err || { tail -20 gpurun_out/r05_i_yahoo.err; exit 1; }
tail -3 gpurun_out/r05_i_yahoo.err; python -c "import json; l=json.load(open('gpurun_out/r05_i_yahoo_line.json')); print(l['ms_per_step'], l['solver_create_s'], l['ndcg10_test'])"

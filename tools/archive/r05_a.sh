#!/bin/bash
# round 5, first GPU call: the -m gpu suite, then the driver's default bench command
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r05_a_gputests.log 2>&1
rc=$?
tail -5 gpurun_out/r05_a_gputests.log
[ $rc -eq 0 ] || exit $rc
python bench.py --full-record gpurun_out/r05_a_bench_full.json > gpurun_out/r05_a_line.json 2> gpurun_out/r05_a_bench.err
rc=$?
tail -20 gpurun_out/r05_a_bench.err
wc -c gpurun_out/r05_a_line.json
exit $rc

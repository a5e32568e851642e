#!/bin/bash
# Dev tool (GPU box): parity tests + ml1m bench + Netflix-shaped run of the working tree, one gpurun call.  r02_ab.sh <tag>
TAG=${1:-ab}
OUT=gpurun_out/$TAG
mkdir -p $OUT
python -m pytest tests/test_gpu_parity.py tests/test_gpu_system.py -m gpu -x -q > $OUT/tests.log 2>&1 || { tail -30 $OUT/tests.log; exit 1; }
tail -2 $OUT/tests.log
python bench.py --no-live-traffic --full-line --no-cpu --no-f64 --verbose > $OUT/bench.json 2> $OUT/bench.err || { tail -20 $OUT/bench.err; exit 1; }
python tools/show_bench.py $OUT/bench.json 2>/dev/null | tail -2 | head -1
grep -E "^  (wall|ustep|spmm|sddmm|vhv|prepare|cg)" $OUT/bench.err | head -20
python tools/run_shape.py --shape netflix -k 100 -t 4 > $OUT/netflix.log 2>&1 || { tail -20 $OUT/netflix.log; exit 1; }
cat $OUT/netflix.log | head -40

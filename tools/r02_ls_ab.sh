#!/bin/bash
# Dev tool (GPU box): lock-step vs per-user U step on three shard sizes.
OUT=gpurun_out/ls_ab; mkdir -p $OUT
for ls in 0 1; do
  python tools/run_shape.py --shape netflix -k 100 -t 4 --tune ustep_lockstep=$ls > $OUT/netflix_ls$ls.log 2>&1 || { tail -20 $OUT/netflix_ls$ls.log; exit 1; }
  echo "== netflix full lockstep=$ls"; grep -E "Iter|wall:ustep|u:|inner|test err" $OUT/netflix_ls$ls.log
  python tools/run_shape.py --shape netflix --d1 48000 --nnz 10000000 -k 100 -t 4 --tune ustep_lockstep=$ls > $OUT/nf10_ls$ls.log 2>&1 || exit 1
  echo "== netflix 10M lockstep=$ls"; grep -E "Iter 4|wall:ustep|inner" $OUT/nf10_ls$ls.log
  python tools/run_shape.py --shape ml1m -k 100 -t 6 --tune ustep_lockstep=$ls > $OUT/ml1m_ls$ls.log 2>&1 || exit 1
  echo "== ml1m lockstep=$ls"; grep -E "Iter 6|wall:ustep|inner" $OUT/ml1m_ls$ls.log
done

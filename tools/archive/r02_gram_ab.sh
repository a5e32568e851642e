#!/bin/bash
# Dev tool (GPU box): dual-form (Gram / MFMA) U-step class on / off, ml1m bench + Netflix-shaped runs.
OUT=gpurun_out/gram_ab; mkdir -p $OUT
for g in 0 128 112 96 64; do
  python tools/run_shape.py --shape ml1m -k 100 -t 8 --tune ustep_gram=$g > $OUT/ml1m_g$g.log 2>&1 || { tail -20 $OUT/ml1m_g$g.log; exit 1; }
  echo "== ml1m gram=$g"; grep -E "Iter 8|wall:ustep|ustep/" $OUT/ml1m_g$g.log | head -12
done
for g in 0 112; do
  python tools/run_shape.py --shape netflix -k 100 -t 4 --tune ustep_gram=$g > $OUT/nf_g$g.log 2>&1 || { tail -20 $OUT/nf_g$g.log; exit 1; }
  echo "== netflix gram=$g"; grep -E "Iter 4|wall:ustep|ustep/" $OUT/nf_g$g.log | head -12
done

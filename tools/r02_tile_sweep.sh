#!/bin/bash
# Dev tool (GPU box): SpMM chunk / SDDMM tile sweep on the Netflix shape (large shards keep 128 / 64 by default).
for kv in "spmm_chunk=128" "spmm_chunk=64" "spmm_chunk=96" "spmm_chunk=192" "spmm_chunk=256" "sddmm_tile=32" "sddmm_tile=96" "sddmm_tile=128" "spmm_tiles=80" "spmm_tiles=320"; do
  echo "== $kv"; python tools/run_shape.py --shape netflix -k 100 -t 3 --tune $kv 2>&1 | grep -E "Iter 3|  sddmm|  spmm "
done

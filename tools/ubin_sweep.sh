#!/bin/bash
# Dev tool: wall time of the U step (ml1m shape) for several U-step class layouts (PCR_UBINS / PCR_USTEP_RESIDENT / PCR_USTEP_SEQ / PCR_CLUSTER_K).
run() { echo "== $1 | top-resident=$2 seq=$3 K=$4"; PCR_UBINS="$1" PCR_USTEP_RESIDENT=$2 PCR_USTEP_SEQ=$3 PCR_CLUSTER_K=$4 timeout -k 10 200 python tools/run_shape.py --shape ${SHAPE:-ml1m} ${SHAPE_ARGS} -k 100 -t 4 2>&1 | grep -E "wall:ustep|Iter 4|ustep/" ; }
run "128:64:0,512:256:0" 1 1 1,8,8
run "128:64:0,512:256:0" 1 0 1,8,8
run "128:64:0,512:256:0" 1 1 4,8,8
run "128:64:0,512:256:0" 1 0 4,8,8
run "128:64:0,512:256:0" 1 0 2,4,4

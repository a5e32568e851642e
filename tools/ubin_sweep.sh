#!/bin/bash
# Dev tool: wall time of the U step for several U-step class layouts (PCR_UBINS / PCR_USTEP_RESIDENT / PCR_USTEP_SEQ / PCR_CLUSTER_K).
# SHAPE / SHAPE_ARGS select the data (default ml1m).
run() { echo "== $1 | top-resident=$2 seq=$3 K=$4"; PCR_UBINS="$1" PCR_USTEP_RESIDENT=$2 PCR_USTEP_SEQ=$3 PCR_CLUSTER_K=$4 timeout -k 10 300 python tools/run_shape.py --shape ${SHAPE:-ml1m} ${SHAPE_ARGS} -k 100 -t 3 2>&1 | grep -E "wall:ustep|Iter 3|ustep/" ; }
run "32:64:1,64:64:1,128:64:0,512:256:0" 1 0 1,4,4
run "32:64:1,64:64:1,128:64:0,512:256:0" 0 0 1,4,4
run "32:64:1,64:64:1,128:64:0,512:256:0,1024:256:0" 1 0 1,4,4
run "32:64:1,64:64:1,128:64:0,256:256:0,512:256:0,1024:256:0" 1 0 1,4,4
run "32:64:1,64:64:1,128:64:1,256:256:0,512:256:0,1024:256:0" 1 0 1,4,4

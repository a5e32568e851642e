#!/bin/bash
# Dev tool (GPU box): U-step class layout knobs on the Netflix shape; prints the U-step wall time.
for kv in "debug=0" "cluster_users=4" "cluster_users=64" "cluster_k=1" "ustep_mode=1" "ubins=32:64:1,64:64:0,128:64:0,256:256:0,512:256:0" "ubins=32:64:1,64:64:0,128:64:0,320:256:0,640:256:0" "ubins=64:64:0,256:64:0,512:256:0" "lanes=6" "lanes=3"; do
  echo "== $kv"; python tools/run_shape.py --shape netflix -k 100 -t 3 --tune "$kv" 2>&1 | grep -E "Iter 3|wall:ustep"
done

#!/bin/bash
# Dev tool (GPU box): ms per outer iteration of the bench workload under a few U-step launch plans (PCR_USTEP_SCHED).
#   bash tools/sched_try.sh "plan1" "plan2" ...     ("" = the default plan)
[ $# -eq 0 ] && set -- "" "2:1,3:2,4:3,0:h,1:0,5:0,6:h" "0:h,1:0,2:1,3:2,4:3,5:0,6:h"
for s in "$@"; do
  for rep in 1 2; do
    PCR_USTEP_SCHED="$s" python bench.py --no-cpu --no-profile --steps 60 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('%-40s %.4f ms' % ('$s', d['ms_per_step']))"
  done
done

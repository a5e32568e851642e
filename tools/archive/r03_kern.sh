#!/bin/bash
# Dev tool (GPU box): per-kernel HIP-event averages of the ml1m bench under launch knobs.  usage: r03_kern.sh "<tune args>" ...
for cfg in "$@"; do
  python3 bench.py --no-live-traffic --no-cpu --no-f64 --no-netflix --full-line --no-rows --steps 40 --warmup 5 --profile-period 4 $cfg 2>/dev/null | python3 -c "
import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); k=d['kernels']
print('%-40s %.4f ms | ' % ('$cfg', d['ms_per_step']) + '  '.join('%s %.1f' % (n, k[n]['avg_us']) for n in ('sddmm','vhv/all','spmm','spmm_fin','cg','prepare/all') if n in k) + '  u_step %.1f' % d['roofline_phase']['u_step']['wall_us_per_step'])"
done

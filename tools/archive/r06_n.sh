#!/bin/bash
# round 6: the complete N > 1 line (ml1m weak leg + Netflix-shaped strong leg) with 4 and 6 ranks on the one GPU (a rehearsal of the
# line's shape and size, not a scaling measurement)
set -o pipefail
mkdir -p gpurun_out/r06_n
for N in 4 6; do
  D=$(python3 -c "print(','.join(['0']*$N))")
  timeout -k 10 500 python bench.py --gpus $N --devices $D --comm p2p --rendezvous gloo --no-cpu --full-record gpurun_out/r06_n/n${N}_full.json > gpurun_out/r06_n/n${N}.out 2> gpurun_out/r06_n/n${N}.err || { echo "N=$N failed"; tail -8 gpurun_out/r06_n/n${N}.err | cut -c1-300; }
  python3 -c "
import json
l=json.loads([x for x in open('gpurun_out/r06_n/n${N}.out').read().split('\n') if x.startswith('{')][-1])
print('N=$N', len(json.dumps(l, separators=(',', ':'))), 'bytes', l['ms_per_step'], l['comm_nranks'], l['config']['exchange'], (l.get('exchange') or {}).get('allreduce_us_avg'))
print('   netflix', l['netflix'].get('ms_per_step'), l['netflix'].get('error'), l['netflix'].get('shards'))"
  grep "pcr\] p2p" gpurun_out/r06_n/n${N}.err | head -2
done

#!/bin/bash
# round 6: two ml1m-size ranks on the one GPU -- this tree against round 5's (build_prev/r5tree), and this tree's knobs
set -o pipefail
mkdir -p gpurun_out/r06_h
C="--gpus 2 --devices 0,0 --comm p2p --rendezvous gloo --no-cpu --no-netflix"
show() { python3 -c "
import json,sys
l=json.loads([x for x in open('$1').read().split('\n') if x.startswith('{')][-1]); ph=l.get('roofline_phase') or {}
print('%-34s %.3f ms  U %s V %s  exchange %s' % ('$2', l['ms_per_step'], (ph.get('u_step') or {}).get('wall_us'), (ph.get('v_step') or {}).get('wall_us'), {k:(l.get('exchange') or {}).get(k) for k in ('allreduce_us_avg','us_per_step')}))"; }
python build_prev/r5tree/bench.py $C --full-record gpurun_out/r06_h/r5_full.json > gpurun_out/r06_h/r5.line 2> gpurun_out/r06_h/r5.err && show gpurun_out/r06_h/r5.line "round 5 tree (lanes=1 by bench)"
python bench.py $C --full-record gpurun_out/r06_h/now_full.json > gpurun_out/r06_h/now.line 2> gpurun_out/r06_h/now.err && show gpurun_out/r06_h/now.line "this tree"
python bench.py $C --tune lanes=4 --full-record gpurun_out/r06_h/now_l4_full.json > gpurun_out/r06_h/now_l4.line 2> gpurun_out/r06_h/now_l4.err && show gpurun_out/r06_h/now_l4.line "this tree, lanes=4"
python bench.py $C --tune p2p_ll=0 --full-record gpurun_out/r06_h/now_host_full.json > gpurun_out/r06_h/now_host.line 2> gpurun_out/r06_h/now_host.err && show gpurun_out/r06_h/now_host.line "this tree, host-synchronised"
python build_prev/r5tree/bench.py $C --tune lanes=4 --full-record gpurun_out/r06_h/r5_l4_full.json > gpurun_out/r06_h/r5_l4.line 2> gpurun_out/r06_h/r5_l4.err && show gpurun_out/r06_h/r5_l4.line "round 5 tree, lanes=4"
python bench.py --no-cpu --no-cli --no-f64 --no-netflix --no-rows --no-live-traffic --no-hbm --full-record gpurun_out/r06_h/one_full.json > gpurun_out/r06_h/one.line 2> gpurun_out/r06_h/one.err && show gpurun_out/r06_h/one.line "one rank, this tree"

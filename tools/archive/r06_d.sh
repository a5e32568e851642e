#!/bin/bash
# round 6, fourth GPU call: the failing-leg test once, the default bench with the HBM side, the Yahoo-shaped share, and an fp64 knob sweep
set -o pipefail
mkdir -p gpurun_out/r06_d
export TMPDIR=/tmp
python -m pytest tests/test_bench_entry.py -x -q -m gpu -k "failing_netflix or two_ranks_on_one_gpu" > gpurun_out/r06_d_tests.$(date +%s).log 2>&1
rc=$?; tail -3 gpurun_out/r06_d_tests.*.log; [ $rc -eq 0 ] || exit $rc
python bench.py --full-record gpurun_out/r06_d_bench_full.json > gpurun_out/r06_d_line.json 2> gpurun_out/r06_d_bench.err || { tail -20 gpurun_out/r06_d_bench.err; exit 1; }
grep -E "traffic|hbm|cli\]" gpurun_out/r06_d_bench.err | tail -8; wc -c gpurun_out/r06_d_line.json
python bench.py --shape yahoo --steps 3 --warmup 1 --no-cpu --no-f64 --full-record gpurun_out/r06_d_yahoo_share_full.json > gpurun_out/r06_d_yahoo_share_line.json 2> gpurun_out/r06_d_yahoo_share.err || { tail -20 gpurun_out/r06_d_yahoo_share.err; exit 1; }
python -c "
import json
for f in ('gpurun_out/r06_d_line.json', 'gpurun_out/r06_d_yahoo_share_line.json'):
    l = json.load(open(f)); print(f, l['ms_per_step'], l['roofline'], l.get('hbm'))
    for k in ('f64', 'netflix'):
        if l.get(k): print('  ', k, l[k].get('ms_per_step'), l[k].get('hbm'))
    print('   cli', (l.get('cli') or {}))
"
# fp64 knob sweep (one box, alternating with the default): ms per step, U-step and V-step walls
B="python bench.py --precision f64 --steps 20 --warmup 5 --no-cpu --no-cli --no-netflix --no-rows --no-live-traffic --no-hbm"
run() { tag=$1; shift; $B "$@" --full-record gpurun_out/r06_d/f64_$tag.json > gpurun_out/r06_d/f64_$tag.line 2> gpurun_out/r06_d/f64_$tag.err || { echo "$tag failed"; tail -3 gpurun_out/r06_d/f64_$tag.err; return; }
  python -c "
import json,sys
l=json.load(open('gpurun_out/r06_d/f64_$tag.line')); ph=l.get('roofline_phase') or {}
print('%-28s %.4f ms (no events %.4f)  U %s  V %s' % ('$tag', l['ms_per_step'], (l.get('ms_per_step')/(1+(l.get('profile_overhead_pct') or 0)/100)), (ph.get('u_step') or {}).get('wall_us'), (ph.get('v_step') or {}).get('wall_us')))"; }
run default_a
run ubins_nores --tune ubins=32:64:0,64:64:0,128:64:0,512:256:0
run ubins_res16 --tune ubins=16:64:1,64:64:0,128:64:0,512:256:0
run ubins_256 --tune ubins=32:64:1,64:64:0,128:64:0,256:256:0,512:256:0
run ubins_coarse --tune ubins=48:64:0,128:64:0,512:256:0
run default_b
run mode1 --tune ustep_mode=1
run mode2 --tune ustep_mode=2
run clu8 --tune cluster_users=8
run clu24 --tune cluster_users=24
run clu32 --tune cluster_users=32
run tiles8 --tune spmm_tiles=8
run tiles16 --tune spmm_tiles=16
run chunk64 --tune spmm_chunk=64
run default_c
run lanes3 --tune lanes=3
run lanes5 --tune lanes=5
run resort16 --tune resort_window=16
run csc --tune sddmm_csc=1

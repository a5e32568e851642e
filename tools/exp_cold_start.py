#!/usr/bin/env python3
"""Dev tool (GPU box): what the first outer iterations cost, one by one: ms, inner counts, and with the sorts' fast path off.
usage: exp_cold_start.py [iterations = 12]"""
import sys, time
sys.path.insert(0, ".")
import torch  # noqa: F401  (one HIP runtime per process: torch first)
import primalcr_amd as pcr
from primalcr_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
R = synth.generate("ml1m")
ds = pcr.Dataset.from_ratings(R)
for knobs in ({}, {"resort_window": 0}):
    with pcr.tuned(**knobs):
        s = pcr.Solver(ds, pcr.Parameter(k=100, do_predict=0, **{"lambda": 5000.0}))
    U0, V0 = pcr.initial(R.d1, 100), pcr.initial(R.d2, 100)
    s.set_factors(U0, V0); s.iterate(1); s.set_factors(U0, V0); s.sync()      # code objects loaded
    s.profile(True, period=1)
    print(knobs or "default")
    for it in range(1, n + 1):
        s.profile_reset(); s.sync()
        t0 = time.perf_counter(); rec = s.iterate(1)[0]; s.sync(); ms = 1e3 * (time.perf_counter() - t0)
        p = s.profile_all()
        avg = lambda k: 1e3 * p[k][0] / max(p[k][1], 1) if k in p else float("nan")
        print(f"  iter {it:2d}: {ms:6.3f} ms  cg_v {rec['cg_v']} ls_v {rec['ls_v']} cg_u {rec['cg_u']:6d} ls_u {rec['ls_u']:5d}  prepare {avg('prepare/all'):6.1f} us  u-step wall {avg('wall:ustep'):6.1f} us")
    s.close()

#!/usr/bin/env python3
"""Dev tool (GPU box): wall time of the evaluator (util.cpp:434-542 on the GPU) on a shape, train and test set."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import primalcr_amd as pcr
from primalcr_amd import synth
shape = sys.argv[1] if len(sys.argv) > 1 else "netflix"
users = (0, int(sys.argv[2])) if len(sys.argv) > 2 else None
R = synth.generate("ml1m") if shape == "ml1m" else synth.generate_fast(shape, users=users)
k = 200 if shape == "yahoo" else 100
s = pcr.Solver(pcr.Dataset.from_ratings(R), pcr.Parameter(k=k, do_predict=0, **{"lambda": 5000.0}))
s.set_factors(pcr.initial(R.d1, k), pcr.initial(R.d2, k))
s.iterate(2)
for which in (0, 1):
    s.evaluate(which, 10)
    s.profile(True, period=1); s.profile_reset()
    t0 = time.perf_counter(); e, n = s.evaluate(which, 10); dt = time.perf_counter() - t0
    p = s.profile_all(); s.profile(False)
    print(f"{shape} {'train' if which == 0 else 'test'} set: pairwise error {e:.6f} ndcg@10 {n:.6f}  {1e3 * dt:.2f} ms wall  " +
          "  ".join(f"{k_} {v[0]:.2f} ms" for k_, v in sorted(p.items()) if k_.startswith(("eval", "wall:eval"))))

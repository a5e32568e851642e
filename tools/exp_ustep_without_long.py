#!/usr/bin/env python3
"""Dev tool (GPU box): what could the U step gain if its LONGEST users cost nothing?  (VERDICT r5 item 4 proposes an explicit-Hessian
truncated CG for the users above 1024 ratings -- the cluster class -- to cut their 22 dependent row passes to three.)  An upper bound
of that gain without writing the kernel: the ml1m bench workload with those users' rating lists cut (a) to 1024 ratings (they join the
513..1024 class), (b) to 32 ratings (they all but vanish), against the full workload -- fork..join wall time of the U step and the
step time, same box, alternating.      python tools/exp_ustep_without_long.py [--precision f32]"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import primalcr_amd as pcr
from primalcr_amd import synth
ap = argparse.ArgumentParser(); ap.add_argument("--precision", default="f32"); a = ap.parse_args()
R = synth.generate("ml1m", seed=synth.SEED)
lens = np.bincount(R.user, minlength=R.d1)
pos = np.arange(R.nnz) - np.repeat(np.concatenate([[0], np.cumsum(lens)[:-1]]), lens)       # position of a rating inside its user


def run(tag, cut):
    keep = np.ones(R.nnz, bool) if cut is None else ~((lens[R.user] > 1024) & (pos >= cut))
    ds = pcr.Dataset.from_triplets(R.d1, R.d2, R.user[keep], R.item[keep], R.val[keep])
    s = pcr.Solver(ds, pcr.Parameter(k=100, precision=pcr.PCR_F32 if a.precision == "f32" else pcr.PCR_F64, do_predict=0, **{"lambda": 5000.0}))
    s.set_factors(pcr.initial(R.d1, 100), pcr.initial(R.d2, 100))
    s.iterate(5)
    s.profile(True, period=4); s.profile_reset()
    import time
    s.sync(); t0 = time.perf_counter(); s.iterate(20); s.sync(); dt = time.perf_counter() - t0
    prof = s.profile_all(); s.profile(False)
    wall = prof.get("wall:ustep", (0, 0))
    cls = {k: round(1e3 * v[0] / v[1], 1) for k, v in prof.items() if k.startswith("ustep/") and v[1]}
    print(f"{tag:44s} {int(keep.sum()):7d} ratings  step {1e3 * dt / 20:.4f} ms  U step fork..join {1e3 * wall[0] / max(wall[1], 1):.1f} us  classes {cls}", flush=True)
    s.close()


n_long = int((lens > 1024).sum())
print(f"ml1m bench workload, {a.precision}: {n_long} users above 1024 ratings hold {int(lens[lens > 1024].sum())} of {R.nnz} ratings")
for rep in range(2):
    run("full workload", None)
    run("long users cut to 1024 ratings", 1024)
    run("long users cut to 32 ratings", 32)

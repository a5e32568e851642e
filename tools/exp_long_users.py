#!/usr/bin/env python3
"""Dev tool: the U step on a shard made ONLY of long users (nusers x n ratings each) -- the per-rating cost of k_ustep's long
classes, to set beside tools/ubench/hess_probe (the cost of building an explicit r x r Hessian for the same users)."""
import argparse, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import primalcr_amd as pcr

ap = argparse.ArgumentParser()
ap.add_argument("--nusers", type=int, default=512); ap.add_argument("-n", type=int, default=4096); ap.add_argument("--d2", type=int, default=17770)
ap.add_argument("-k", type=int, default=100); ap.add_argument("-t", type=int, default=4); ap.add_argument("-l", type=float, default=5000.0)
ap.add_argument("--tune", action="append", default=[])
a = ap.parse_args()
for kv in a.tune:
    pcr.tune(*kv.split("=", 1))
rng = np.random.default_rng(3)
index = np.arange(a.nusers + 1, dtype=np.int64) * a.n
item = np.concatenate([np.sort(rng.choice(a.d2, a.n, replace=False)) for _ in range(a.nusers)]).astype(np.int32)
# ratings 1..5 correlated with a low-rank score, as the generator of the shapes does
zu = rng.standard_normal((a.nusers, 8)); zi = rng.standard_normal((a.d2, 8))
sc = np.einsum("ij,ij->i", np.repeat(zu, a.n, axis=0), zi[item]) / np.sqrt(8) + 0.7 * rng.standard_normal(item.size)
val = np.clip(np.round(3.5 + 1.1 * sc), 1, 5).astype(np.float64)
tindex = np.arange(a.nusers + 1, dtype=np.int64) * 2
titem = np.tile(np.array([0, 1], np.int32), a.nusers); tval = np.tile(np.array([1.0, 2.0]), a.nusers)
ds = pcr.Dataset.from_csr(a.nusers, a.d2, index, item, val, tindex, titem, tval)
p = pcr.Parameter(k=a.k, maxiter=a.t, do_predict=0, solver_type=2, **{"lambda": a.l})
s = pcr.Solver(ds, p)
s.set_factors(pcr.initial(a.nusers, a.k), pcr.initial(a.d2, a.k))
s.profile(True, period=1)
recs, _ = s.train(log=lambda l: print("  gpu|", l, flush=True))
prof = s.profile_all()
nr = a.nusers * a.n
for kname, (ms, cnt) in sorted(prof.items(), key=lambda kv: -kv[1][0])[:12]:
    print(f"  {kname:14s} {ms:10.2f} ms {cnt:6d} timed  {1e3*ms/max(cnt,1):10.1f} us/launch  {1e6*ms/max(cnt,1)/nr:7.3f} ns per rating")
print("inner counts (cg_v, ls_v, cg_u, ls_u):", [(r["cg_v"], r["ls_v"], r["cg_u"], r["ls_u"]) for r in recs[1:]])

#!/usr/bin/env python3
"""Dev tool: fraction of ratings that have at least one ACTIVE partner (a rating of another level of the same user
inside the hinge margin) after t outer iterations -- the ratings whose sweep coefficient can be non-zero."""
import argparse, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import primalcr_amd as pcr
from primalcr_amd import synth
ap = argparse.ArgumentParser(); ap.add_argument("--shape", default="ml1m"); ap.add_argument("-t", type=int, default=5); ap.add_argument("-k", type=int, default=100)
ap.add_argument("--d1", type=int); ap.add_argument("--nnz", type=int)
a = ap.parse_args()
R = synth.generate(a.shape, d1=a.d1, nnz=a.nnz)
ds = pcr.Dataset.from_ratings(R)
s = pcr.Solver(ds, pcr.Parameter(k=a.k, **{"lambda": 5000.0}))
s.set_factors(pcr.initial(R.d1, a.k), pcr.initial(R.d2, a.k))
idx, item, val = ds.csr(0)
user = np.repeat(np.arange(R.d1), np.diff(idx))
lev = np.rint(val).astype(np.int64); T = int(lev.max()) + 1
for it in range(a.t + 1):
    m = s.comp_m()
    lo = np.full((R.d1, T), np.inf); hi = np.full((R.d1, T), -np.inf)
    np.minimum.at(lo, (user, lev), m); np.maximum.at(hi, (user, lev), m)
    # min m over higher levels / max m over lower levels, per (user, level)
    min_above = np.full((R.d1, T), np.inf); max_below = np.full((R.d1, T), -np.inf)
    for l in range(T - 2, -1, -1): min_above[:, l] = np.minimum(min_above[:, l + 1], lo[:, l + 1])
    for l in range(1, T): max_below[:, l] = np.maximum(max_below[:, l - 1], hi[:, l - 1])
    act = (min_above[user, lev] <= m + 1) | (max_below[user, lev] >= m - 1)
    print(f"iter {it}: active ratings {act.mean():.3f}", flush=True)
    if it < a.t: s.iterate(1)

#!/usr/bin/env python3
"""Dev tool: find the users on which the GPU evaluator and the oracle disagree (tests/golden/toy_test after one iteration)."""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import primalcr_amd as pcr
from oracle import oracle_py
g = np.load(os.path.join(ROOT, "tests/golden/toy_test.npz")); meta = json.load(open(os.path.join(ROOT, "tests/golden/toy_test.json")))
u, i, v = g["user"].astype(np.int64), g["item"].astype(np.int64), g["val"].astype(np.float64)
orc = oracle_py.Oracle()
d1, d2, k, lam = meta["d1"], meta["d2"], 10, 5000.0
X = orc.build_csr(d1, d2, u, i, v)
U = orc.initial(d1, k); V = orc.initial(d2, k)
V1, mm, _, _ = orc.update_V_new(X, lam, 1.0, U, V)
U1, _, _ = orc.update_U_new(X, mm, lam, 1.0, V1, U)
def both(lo, hi):
    keep = (u >= lo) & (u < hi)
    uu, ii, vv = u[keep] - lo, i[keep], v[keep]
    Xs = orc.build_csr(hi - lo, d2, uu, ii, vv)
    s = pcr.Solver(pcr.Dataset.from_triplets(hi - lo, d2, uu.astype(np.int32), ii.astype(np.int32), vv), pcr.Parameter(k=k, precision=pcr.PCR_F64))
    s.set_factors(U1[lo:hi], V1)
    return s.evaluate(0), orc.eval(U1[lo:hi], V1, Xs)
lo, hi = 0, d1
print("all", both(lo, hi))
while hi - lo > 1:
    mid = (lo + hi) // 2
    a, b = both(lo, mid)
    if abs(a[0] - b[0]) > 1e-12 or abs(a[1] - b[1]) > 1e-9: hi = mid
    else: lo = mid
print("first bad user", lo, both(lo, lo + 1))
sel = u == lo
sc = V1[i[sel]] @ U1[lo]
print("n", sel.sum(), "u norm", np.abs(U1[lo]).max(), "vals", v[sel][:8], "scores", sc[:8], "distinct vals", len(set(v[sel])), "ties", len(sc) - len(set(sc)))

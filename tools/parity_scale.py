#!/usr/bin/env python3
"""Dev tool: half-step by half-step parity of the GPU path (fp64) against the C oracle on a larger synthetic shape
(long-tailed users: every length class, clusters, global-scratch users).  Prints the relative differences of V, U and
the objectives after every V step and U step."""
import argparse, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import primalcr_amd as pcr
from primalcr_amd import synth
from oracle import oracle_py

ap = argparse.ArgumentParser()
ap.add_argument("--shape", default="netflix"); ap.add_argument("--d1", type=int, default=8000); ap.add_argument("--nnz", type=int, default=1600000)
ap.add_argument("-k", type=int, default=16); ap.add_argument("-t", type=int, default=3); ap.add_argument("-l", type=float, default=5000.0)
ap.add_argument("--f32", action="store_true")
a = ap.parse_args()
R = synth.generate(a.shape, d1=a.d1, nnz=a.nnz)
orc = oracle_py.Oracle()
X = orc.build_csr(R.d1, R.d2, R.user, R.item, R.val)
ds = pcr.Dataset.from_ratings(R)
lens = np.diff(ds.csr(0)[0])
print(f"[data] {R.d1}x{R.d2} nnz={R.nnz} max len {lens.max()} >4096: {(lens>4096).sum()} >1024: {(lens>1024).sum()}", flush=True)
U = orc.initial(R.d1, a.k); V = orc.initial(R.d2, a.k)
s = pcr.Solver(ds, pcr.Parameter(k=a.k, precision=pcr.PCR_F32 if a.f32 else pcr.PCR_F64, **{"lambda": a.l}))
s.set_factors(U, V)
rel = lambda x, y: float(np.abs(x - y).max() / max(np.abs(y).max(), 1e-300))
for it in range(1, a.t + 1):
    t0 = time.time()
    V, m, objV, iv = orc.update_V_new(X, a.l, 1.0, U, V)
    gV, giv = s.update_V()
    Ug, Vg = s.get_factors()
    print(f"iter {it} V step: obj rel {abs(gV/objV-1):.2e}  V rel {rel(Vg, V):.2e}  cg/ls oracle {iv['cg']}/{iv['ls']} gpu {giv['cg']}/{giv['ls']}  ({time.time()-t0:.1f}s)", flush=True)
    t0 = time.time()
    U, objU, iu = orc.update_U_new(X, m, a.l, 1.0, V, U)
    gU, giu = s.update_U()
    Ug, Vg = s.get_factors()
    d = np.abs(Ug - U).max(axis=1) / max(np.abs(U).max(), 1e-300)
    worst = np.argsort(-d)[:5]
    print(f"iter {it} U step: obj rel {abs(gU/objU-1):.2e}  U rel {d.max():.2e}  cg/ls oracle {iu['cg']}/{iu['ls']} gpu {giu['cg']}/{giu['ls']}  "
          f"worst users {[(int(u), int(lens[u]), float(d[u])) for u in worst]}  ({time.time()-t0:.1f}s)", flush=True)

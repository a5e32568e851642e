#!/usr/bin/env python3
"""Dev tool: train on a synthetic shape on the GPU, print per-kernel timings, optionally compare the
objective / NDCG trajectory with the reference binary (oracle/_ref/omp-pmf-train) on the same data."""
import argparse, os, re, subprocess, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import primalcr_amd as pcr
from primalcr_amd import synth

ap = argparse.ArgumentParser()
ap.add_argument("--shape", default="ml1m"); ap.add_argument("--d1", type=int); ap.add_argument("--d2", type=int)
ap.add_argument("--nnz", type=int); ap.add_argument("--mu", type=float); ap.add_argument("--sigma", type=float)
ap.add_argument("-k", type=int, default=100); ap.add_argument("-t", type=int, default=3); ap.add_argument("-l", type=float, default=5000.0)
ap.add_argument("--f64", action="store_true"); ap.add_argument("--ref", action="store_true"); ap.add_argument("--predict", type=int, default=0)
ap.add_argument("--threads", type=int, default=16); ap.add_argument("-s", type=int, default=2, help="1 PrimalCR, 2 PrimalCR++")
ap.add_argument("--cg-iters", type=int, default=10); ap.add_argument("--cg-tol", type=float, default=0.01)
ap.add_argument("--tune", action="append", default=[], help="key=value launch knob (pcr_tune), repeatable")
ap.add_argument("--numpy-gen", action="store_true", help="the numpy generator for every shape (round 1's Netflix-shaped set: minutes)")
a = ap.parse_args()
for kv in a.tune:
    pcr.tune(*kv.split("=", 1))
t0 = time.time()
ap_users = None
if a.shape in ("netflix", "yahoo") and not a.numpy_gen:      # the C++ generator (seconds); --d1 N on yahoo = its first N users
    R = synth.generate_fast(a.shape, users=(0, a.d1) if (a.shape == "yahoo" and a.d1) else None, d1=None if a.shape == "yahoo" else a.d1, d2=a.d2, nnz=a.nnz, mu=a.mu, sigma=a.sigma)
else:
    R = synth.generate(a.shape, d1=a.d1, d2=a.d2, nnz=a.nnz, mu=a.mu, sigma=a.sigma)
ds = pcr.Dataset.from_ratings(R)
idx, _, _ = ds.csr(0); lens = np.diff(idx)
print(f"[data] {R.d1}x{R.d2} nnz={R.nnz} pairs={ds.count_pairs()} len max={lens.max()} mean={lens.mean():.1f} >4096: {(lens>4096).sum()} ({time.time()-t0:.1f}s)", flush=True)
p = pcr.Parameter(k=a.k, maxiter=a.t, do_predict=a.predict, solver_type=a.s, cg_max_iter=a.cg_iters, cg_tol=a.cg_tol, precision=pcr.PCR_F64 if a.f64 else pcr.PCR_F32, **{"lambda": a.l})
s = pcr.Solver(ds, p)
s.set_factors(pcr.initial(R.d1, a.k), pcr.initial(R.d2, a.k))
s.profile(True, period=2)
recs, lines = s.train(log=lambda l: print("  gpu|", l, flush=True))
prof = s.profile_all(); tot = sum(v[0] for v in prof.values())
for kname, (ms, n) in sorted(prof.items(), key=lambda kv: -kv[1][0])[:30]:
    print(f"  {kname:14s} {ms:10.2f} ms {n:6d} timed  {1e3*ms/max(n,1):10.1f} us/launch {100*ms/tot:5.1f}%")
print("inner counts:", [(r["cg_v"], r["ls_v"], r["cg_u"], r["ls_u"]) for r in recs[1:]])
te = s.evaluate(1, 10); print("test err/ndcg", te, flush=True)
if a.ref:
    from oracle import oracle_py
    with tempfile.TemporaryDirectory() as td:
        d = synth.write_dir(R, os.path.join(td, "data"))
        t0 = time.time()
        out = subprocess.run([oracle_py.REF_TRAIN, "-k", str(a.k), "-l", repr(a.l), "-t", str(a.t), "-p", str(a.predict), "-n", str(a.threads), d, os.path.join(td, "m")],
                             cwd=td, capture_output=True, text=True, check=True).stdout
        print(f"[ref] wall {time.time()-t0:.1f}s")
    for l in out.split("\n"):
        if l.startswith(("Iter", "(T")): print("  ref|", l)
    ro = [float(x) for x in re.findall(r"^Iter \d+ time \S+ obj (\S+)$", out, re.M)]
    print("objective rel diff per iter:", [abs(r["obj"] / o - 1) for r, o in zip(recs, ro)])

#!/usr/bin/env python3
"""Dev tool: re-run one case of tests/test_gpu_parity.py::test_fuzz_small_shapes_against_oracle step by step against the oracle."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import primalcr_amd as pcr
from oracle.oracle_py import Oracle
want = int(sys.argv[1]) if len(sys.argv) > 1 else 17
oracle = Oracle()
rng = np.random.default_rng(2026)
rel = lambda a, b: float(np.abs(np.asarray(a) - np.asarray(b)).max() / max(np.abs(np.asarray(b)).max(), 1e-300))
for case in range(80):
    d1 = int(rng.integers(3, 60)); d2 = int(rng.integers(20, 900))
    r = int(rng.choice([1, 2, 3, 5, 7, 8, 12, 17, 33, 64]))
    nlev = int(rng.choice([1, 2, 3, 5, 9, 10, 12]))
    solver = int(rng.choice([1, 2]))
    real = bool(rng.integers(0, 2)) and nlev > 1
    lam = float(rng.choice([0.5, 5.0, 50.0, 500.0]))
    lens = np.minimum(rng.choice([0, 1, 2, 3, 10, 40, 64, 65, 130, 256, 257, 300, 700], d1), d2)
    lens[rng.integers(0, d1)] = min(d2, 64)
    user = np.repeat(np.arange(d1), lens)
    item = np.concatenate([rng.choice(d2, n, replace=False) for n in lens]) if user.size else np.zeros(0, np.int64)
    val = rng.integers(1, nlev + 1, user.shape[0]).astype(np.float64)
    if real:
        val = val + rng.uniform(-0.49, 0.49, val.shape[0])
    if user.size == 0:
        continue
    if case != want:
        continue
    print(dict(case=case, d1=d1, d2=d2, r=r, nlev=nlev, solver=solver, real=real, lam=lam, nnz=int(user.size)), flush=True)
    X = oracle.build_csr(d1, d2, user, item, val)
    U0 = oracle.initial(d1, r) * 0.4; V0 = oracle.initial(d2, r) * 0.4
    s = pcr.Solver(pcr.Dataset.from_triplets(d1, d2, user, item, val), pcr.Parameter(k=r, solver_type=solver, precision=pcr.PCR_F64, **{"lambda": lam}))
    s.set_factors(U0, V0)
    U, V = U0, V0
    for it in range(2):
        mo = oracle.comp_m(U, V, X)
        print(" m", rel(s.comp_m(), mo), "obj", s.objective() / oracle.objective_new(mo, U, V, X, lam, solver=solver) - 1)
        g = s.obtain_g(); go = oracle.obtain_g_new(U, V, X, mo, lam, solver=solver)
        print(" g", rel(g, go))
        a = np.random.default_rng(it).normal(size=V.shape)
        print(" Ha", rel(s.compute_Ha(a), oracle.compute_Ha_new(a, mo, U, X, lam, solver=solver)))
        s.set_factors(U, V)
        oV, iv = s.update_V(); oU, iu = s.update_U()
        if solver == 2:
            V1, m1, objVo, ivo = oracle.update_V_new(X, lam, 1.0, U, V)
            U1, objUo, iuo = oracle.update_U_new(X, m1, lam, 1.0, V1, U)
            print(" V step", oV / objVo - 1, iv, ivo, " U step", oU / objUo - 1, iu, iuo)
        Ug, Vg = s.get_factors()
        if solver == 2:
            print(" V", rel(Vg, V1), "U", rel(Ug, U1))
        U, V = Ug, Vg
    s.set_factors(U0, V0)
    got = s.iterate(2)
    Uo, Vo, recs = oracle.train(X, U0, V0, lam, 2, solver=solver, do_predict=0)
    for gi, oi in zip(got, recs[1:]):
        print(" iterate obj %.12f oracle %.12f rel %.2e counts gpu %s oracle %s" % (gi["obj"], oi["obj"], gi["obj"] / oi["obj"] - 1,
              (gi["cg_v"], gi["ls_v"], gi["cg_u"], gi["ls_u"]), (oi["cg_v"], oi["ls_v"], oi["cg_u"], oi["ls_u"])))

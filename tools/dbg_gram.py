#!/usr/bin/env python3
"""Dev tool: replay a case of test_fuzz_small_shapes_against_oracle under pcr_tune knobs and show which users differ."""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import primalcr_amd as pcr
from oracle.oracle_py import Oracle
want = int(sys.argv[1]); knobs = dict(kv.split("=") for kv in sys.argv[2:])
oracle = Oracle()
rng = np.random.default_rng(2026)
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
    if user.size == 0 or case != want:
        continue
    print(dict(case=case, d1=d1, d2=d2, r=r, nlev=nlev, solver=solver, real=real, lam=lam), "lens", lens.tolist())
    X = oracle.build_csr(d1, d2, user, item, val)
    U0 = oracle.initial(d1, r) * 0.4; V0 = oracle.initial(d2, r) * 0.4
    V1, m1, objV, iv = (oracle.update_V_new(X, lam, 1.0, U0, V0) if solver == 2 else (None,) * 4)
    with pcr.tuned(**knobs):
        s = pcr.Solver(pcr.Dataset.from_triplets(d1, d2, user, item, val), pcr.Parameter(k=r, solver_type=solver, precision=pcr.PCR_F64, **{"lambda": lam}))
    s.set_factors(U0, V0)
    print("V step", s.update_V(), objV, iv)
    if solver == 2:
        U1, objU, iu = oracle.update_U_new(X, m1, lam, 1.0, V1, U0)
        print("U step", s.update_U(), objU, iu)
        Ug, _ = s.get_factors()
        bad = np.flatnonzero(~np.isfinite(Ug).all(1) | (np.abs(Ug - U1).max(1) > 1e-7 * max(1e-3, np.abs(U1).max())))
        print("bad users", bad.tolist(), "their lengths", lens[bad].tolist())
        for u in bad[:3]:
            print(u, Ug[u], U1[u])
    else:
        print("U step", s.update_U())
    for rep in range(4):
        with pcr.tuned(**knobs):
            s2 = pcr.Solver(pcr.Dataset.from_triplets(d1, d2, user, item, val), pcr.Parameter(k=r, solver_type=solver, precision=pcr.PCR_F64, **{"lambda": lam}))
        s2.set_factors(U0, V0)
        got = s2.iterate(2)
        Ug, Vg = s2.get_factors()
        print("iterate:", [(g["obj"], g["cg_u"], g["ls_u"]) for g in got], "finite U", np.isfinite(Ug).all(), "finite V", np.isfinite(Vg).all(),
              "nan users", np.flatnonzero(~np.isfinite(Ug).all(1)).tolist())
    with pcr.tuned(**knobs):
        s3 = pcr.Solver(pcr.Dataset.from_triplets(d1, d2, user, item, val), pcr.Parameter(k=r, solver_type=solver, precision=pcr.PCR_F64, **{"lambda": lam}))
    s3.set_factors(U0, V0)
    for it in range(2):
        print("  V", s3.update_V())
        Ub, Vb = s3.get_factors()
        print("  U", s3.update_U())
        Ua, Va = s3.get_factors()
        nanu = np.flatnonzero(~np.isfinite(Ua).all(1))
        print("   |U before| max", np.abs(Ub).max(), "min row norm", np.sqrt((Ub**2).sum(1)).min(), "nan users", nanu.tolist()[:8])
        if nanu.size:
            u = nanu[0]
            print("   user", u, "len", lens[u], "u before", Ub[u][:4], "V rows max", np.abs(Vb[item[user == u]]).max())
    with pcr.tuned(**knobs):
        s4 = pcr.Solver(pcr.Dataset.from_triplets(d1, d2, user, item, val), pcr.Parameter(k=r, solver_type=solver, precision=pcr.PCR_F64, **{"lambda": lam}))
    s4.set_factors(U0, V0)
    s4.update_V(); s4.update_U()
    print("objective from the state the U step left:", s4.objective())
    g = s4.obtain_g(); print("g finite", np.isfinite(g).all(), "rows with nan", np.flatnonzero(~np.isfinite(g).all(1))[:10].tolist())
    Ua, Va = s4.get_factors()
    s4.set_factors(Ua, Va); m = s4.comp_m(); print("fresh state objective", s4.objective(), "m finite", np.isfinite(m).all())
    g2 = s4.obtain_g(); print("fresh g finite", np.isfinite(g2).all())
    with pcr.tuned(**knobs):
        s5 = pcr.Solver(pcr.Dataset.from_triplets(d1, d2, user, item, val), pcr.Parameter(k=r, solver_type=solver, precision=pcr.PCR_F64, **{"lambda": lam}))
    s5.set_factors(U0, V0)
    s5.update_V(); s5.update_U()
    a = np.random.default_rng(0).normal(size=V0.shape)
    Ha = s5.compute_Ha(a); print("Ha finite", np.isfinite(Ha).all(), "nan rows", np.flatnonzero(~np.isfinite(Ha).all(1))[:10].tolist())
    g = s5.obtain_g(); d, its = s5.solve_delta(g); print("delta finite", np.isfinite(d).all(), its)
    Ua, Va = s5.get_factors(); print("U finite", np.isfinite(Ua).all(), "V finite", np.isfinite(Va).all(), "|V|max", np.abs(Va).max())

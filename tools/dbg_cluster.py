import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import primalcr_amd as pcr
from oracle.oracle_py import Oracle
oracle = Oracle()
prec = pcr.PCR_F64 if len(sys.argv) < 2 or sys.argv[1] == "f64" else pcr.PCR_F32
rng = np.random.default_rng(5)
d1, d2, r, lam = 40, 6000, 12, 30.0
lens = np.concatenate([[0, 1, 2, 5000, 4097, 4096, 1500, 1024, 1025, 300, 256, 257, 64, 65], rng.integers(3, 200, d1 - 14)])
user = np.repeat(np.arange(d1), lens)
item = np.concatenate([rng.choice(d2, n, replace=False) for n in lens])
val = rng.integers(1, 6, user.shape[0]).astype(np.float64)
X = oracle.build_csr(d1, d2, user, item, val)
U = oracle.initial(d1, r) * 0.3; V = oracle.initial(d2, r) * 0.3
ds = pcr.Dataset.from_triplets(d1, d2, user, item, val)
s = pcr.Solver(ds, pcr.Parameter(k=r, precision=prec, **{"lambda": lam}))
s.set_factors(U, V)
s.comp_m(want=False)
mo = oracle.comp_m(U, V, X)
Uo, objo, info_o = oracle.update_U_new(X, mo, lam, 1.0, V, U)
objU, info = s.update_U()
Ug, _ = s.get_factors()
print("obj", objU, objo, info, info_o)
for i in range(d1):
    e = np.abs(Ug[i]-Uo[i]).max()/max(np.abs(Uo[i]).max(),1e-30)
    if e > 1e-9 or lens[i] > 1000: print(i, lens[i], e)

import sys, time
sys.path.insert(0, "/root/repo")
import primalcr_amd as pcr
from primalcr_amd import synth
t=time.time(); R = synth.generate_fast("netflix"); print("gen", time.time()-t)
t=time.time(); ds = pcr.Dataset.from_ratings(R); print("dataset", time.time()-t)
t=time.time(); n=ds.count_pairs(); print("count_pairs", time.time()-t)
pcr.tune("debug", 1)
for tag in ("debug",):
    t=time.time(); s = pcr.Solver(ds, pcr.Parameter(k=100, do_predict=0, **{"lambda": 5000.0})); print("create", time.time()-t)
t=time.time(); U=pcr.initial(R.d1,100); V=pcr.initial(R.d2,100); print("initial", time.time()-t)
t=time.time(); s.set_factors(U,V); print("set_factors", time.time()-t)
t=time.time(); s.iterate(1); print("first iter", time.time()-t)
t=time.time(); s.iterate(2); print("2 iters", time.time()-t)

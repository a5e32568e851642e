import sys, os, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import primalcr_amd as pcr
from primalcr_amd import synth
R = synth.generate("ml1m"); ds = pcr.Dataset.from_ratings(R)
s = pcr.Solver(ds, pcr.Parameter(k=100, **{"lambda": 5000.0}))
s.set_factors(pcr.initial(R.d1,100), pcr.initial(R.d2,100))
for _ in range(2): s.update_V(); s.update_U()
print("train", s.evaluate(0), "test", s.evaluate(1))
s.sync(); t=time.perf_counter()
for _ in range(10): s.evaluate(0); s.evaluate(1)
s.sync(); print(os.environ.get("PCR_EVAL_BRUTE","0"), "ms per (train+test) eval:", 1e2*(time.perf_counter()-t))

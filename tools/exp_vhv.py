import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import primalcr_amd as pcr
from primalcr_amd import synth
R = synth.generate("ml1m"); ds = pcr.Dataset.from_ratings(R)
s = pcr.Solver(ds, pcr.Parameter(k=100, **{"lambda": 5000.0}))
s.set_factors(pcr.initial(R.d1,100), pcr.initial(R.d2,100))
s.comp_m(want=False)
a = np.random.default_rng(0).normal(size=(R.d2,100))
s.compute_Ha(a)
s.profile(True); s.profile_reset()
for _ in range(20): s.compute_Ha(a)
tag = " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith("PCR_"))
print(tag, " ".join(f"{k}:{1e3*ms/n:.1f}" for k,(ms,n) in sorted(s.profile_all().items())))

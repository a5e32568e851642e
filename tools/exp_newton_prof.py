#!/usr/bin/env python3
"""Dev tool (GPU box): in-kernel phase clocks of k_unewton (a -DPCR_NEWTON_PROF build in build_next/nprof) on the ml1m bench workload:
    make -C primalcr_amd/csrc lib LIBDIR=$PWD/build_next/nprof PCR_EXTRA=-DPCR_NEWTON_PROF;  python tools/exp_newton_prof.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import primalcr_amd as pcr
from primalcr_amd import synth
pcr.use_library(os.path.join(ROOT, "build_next", "nprof", "libprimalcr.so"))
R = synth.generate("ml1m", seed=synth.SEED)
ds = pcr.Dataset.from_ratings(R)
with pcr.tuned(ustep_newton=1):
    s = pcr.Solver(ds, pcr.Parameter(k=100, precision=pcr.PCR_F32, do_predict=0, **{"lambda": 5000.0}))
s.set_factors(pcr.initial(R.d1, 100), pcr.initial(R.d2, 100))
s.iterate(3)
s.sync()

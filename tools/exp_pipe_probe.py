#!/usr/bin/env python3
"""Dev tool (GPU box): TIMING of one V-side Hessian-vector product as it runs today against its two-stream form.
Needs the probe build:  make -C primalcr_amd/csrc lib LIBDIR=$PWD/build_next/probe PCR_EXTRA=-DPCR_PIPE_PROBE
    python tools/exp_pipe_probe.py [--precision f32|f64] [--shape ml1m|netflix]
The probe (Solver::pipe_probe, pcr_solver.hip) prints its table on stderr from inside pcr_compute_Ha."""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import primalcr_amd as pcr
from primalcr_amd import synth
ap = argparse.ArgumentParser()
ap.add_argument("--precision", default="f32"); ap.add_argument("--shape", default="ml1m")
a = ap.parse_args()
pcr.use_library(os.path.join(ROOT, "build_next", "probe", "libprimalcr.so"))
if a.shape == "ml1m":
    R = synth.generate("ml1m", seed=synth.SEED)
else:
    R = synth.generate_fast("netflix", d1=48000, nnz=10_000_000)
ds = pcr.Dataset.from_ratings(R)
s = pcr.Solver(ds, pcr.Parameter(k=100, precision=pcr.PCR_F32 if a.precision == "f32" else pcr.PCR_F64, do_predict=0, **{"lambda": 5000.0}))
s.set_factors(pcr.initial(R.d1, 100), pcr.initial(R.d2, 100))
s.iterate(6)
print(f"[pipe-probe] {a.shape} {a.precision}: {R.d1} users, {R.nnz} ratings", file=sys.stderr)
s.compute_Ha(pcr.initial(R.d2, 100) * 0.1)

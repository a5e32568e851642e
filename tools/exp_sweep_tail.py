#!/usr/bin/env python3
"""Dev tool (GPU box): how much of k_vsweep_all / k_prepare_all is the chain of the longest users?  Per-kernel times on the ml1m
shape with the NB longest users removed."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import primalcr_amd as pcr
from primalcr_amd import synth
R = synth.generate("ml1m")
ds = pcr.Dataset.from_ratings(R)
idx, item, val = ds.csr(0)
lens = np.diff(idx)
order = np.argsort(-lens, kind="stable")
for nb in (0, 8, 32, 128, 512):
    mask = np.ones(R.d1, bool); mask[order[:nb]] = False
    keep = np.repeat(mask, lens)
    nidx = np.concatenate([[0], np.cumsum(lens[mask])]).astype(np.int64)
    d = pcr.Dataset.from_csr(int(mask.sum()), R.d2, nidx, item[keep].astype(np.int32), val[keep])
    s = pcr.Solver(d, pcr.Parameter(k=100, do_predict=0, **{"lambda": 5000.0}))
    s.set_factors(pcr.initial(int(mask.sum()), 100), pcr.initial(R.d2, 100))
    s.iterate(3)
    s.profile(True, period=1); s.profile_reset()
    s.iterate(6)
    p = s.profile_all()
    avg = lambda k: 1e3 * p[k][0] / max(p[k][1], 1) if k in p else float("nan")
    print(f"without the {nb:3d} longest users (longest left {lens[mask].max():4d}, {int(lens[mask].sum())} ratings): vhv/all {avg('vhv/all'):6.1f} us  "
          f"prepare/all {avg('prepare/all'):6.1f} us  sddmm {avg('sddmm'):5.1f}  spmm {avg('spmm'):5.1f}  wall:ustep {avg('wall:ustep'):6.1f}")
    s.close()

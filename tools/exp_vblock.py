#!/usr/bin/env python3
"""Dev tool (GPU box): the blocked-user MFMA V-step question answered with numbers from this chip.
For the NB longest users of a shape: what do their ratings cost inside the product's k_sddmm / k_spmm launches (full set minus
the set without them; and the block as a data set of its own), against the same block as dense fp32 MFMA GEMMs
(tools/ubench/vblock_probe)?   exp_vblock.py --shape ml1m|netflix --nb 256"""
import argparse, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import primalcr_amd as pcr
from primalcr_amd import synth

ap = argparse.ArgumentParser()
ap.add_argument("--shape", default="ml1m"); ap.add_argument("--nb", type=int, default=256); ap.add_argument("-k", type=int, default=100)
a = ap.parse_args()
R = synth.generate("ml1m") if a.shape == "ml1m" else synth.generate_fast(a.shape)
ds = pcr.Dataset.from_ratings(R)
idx, item, val = ds.csr(0)
lens = np.diff(idx)
top = np.sort(np.argsort(-lens, kind="stable")[:a.nb])
is_top = np.zeros(R.d1, bool); is_top[top] = True


def subset(mask):
    keep = np.repeat(mask, lens)
    nidx = np.concatenate([[0], np.cumsum(lens[mask])]).astype(np.int64)
    return pcr.Dataset.from_csr(int(mask.sum()), R.d2, nidx, item[keep].astype(np.int32), val[keep])


def gather_times(dset, d1):
    s = pcr.Solver(dset, pcr.Parameter(k=a.k, do_predict=0, **{"lambda": 5000.0}))
    s.set_factors(pcr.initial(d1, a.k), pcr.initial(R.d2, a.k))
    recs = s.iterate(2)                                      # a model two iterations in, as in the bench's timed region
    s.comp_m(want=False)
    avec = np.random.default_rng(5).normal(size=(R.d2, a.k))
    s.compute_Ha(avec)
    s.profile(True, period=1); s.profile_reset()
    for _ in range(6):                                       # one k_sddmm + sweep + k_spmm + k_spmm_fin each, none of them skipped
        s.compute_Ha(avec)
    p = s.profile_all()
    out = {k: 1e3 * p[k][0] / max(p[k][1], 1) for k in ("sddmm", "spmm", "spmm_fin")}
    s.close()
    return out

nz_top = int(lens[top].sum())
print(f"{a.shape}: {R.d1} users x {R.d2} items, {R.nnz} ratings; the {a.nb} longest users hold {nz_top} ratings "
      f"({100 * nz_top / R.nnz:.1f} %), block density {100 * nz_top / (a.nb * R.d2):.1f} %, shortest of them {lens[top].min()}")
full = gather_times(ds, R.d1)
rest = gather_times(subset(~is_top), R.d1 - a.nb)
blk = gather_times(subset(is_top), a.nb)
for k in ("sddmm", "spmm", "spmm_fin"):
    print(f"  {k:9s} full set {full[k]:9.2f} us   without the block {rest[k]:9.2f} us   marginal {full[k] - rest[k]:8.2f} us "
          f"= {1e3 * (full[k] - rest[k]) / nz_top:.4f} ns per rating   the block alone {blk[k]:8.2f} us")
cf = "/tmp/vblock_counts.txt"
np.savetxt(cf, lens[top], fmt="%d")
print(subprocess.run([os.path.join(ROOT, "tools", "ubench", "_build", "vblock_probe"), cf, str(R.d2), str(a.k)], capture_output=True, text=True).stdout)

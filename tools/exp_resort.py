#!/usr/bin/env python3
"""Dev tool (GPU box): how far does a user's (level, m) order move between two consecutive sorted states?  (VERDICT r02 item 4-i:
the sorts of k_prepare and of k_ustep's line search run the full bitonic network every time.)
For every half step of T outer iterations: the fraction of users whose PREVIOUS permutation still sorts the new scores, the
fraction of adjacent pairs of the previous order that are now inverted, and the largest rank displacement -- rating-weighted
too, because the long users are the ones whose sort takes time."""
import argparse, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import primalcr_amd as pcr
from primalcr_amd import synth

ap = argparse.ArgumentParser()
ap.add_argument("--shape", default="ml1m"); ap.add_argument("--users", type=int); ap.add_argument("--nnz", type=int)
ap.add_argument("-t", type=int, default=25); ap.add_argument("-k", type=int, default=100)
a = ap.parse_args()
R = synth.generate("ml1m") if a.shape == "ml1m" else synth.generate_fast(a.shape, d1=a.users, nnz=a.nnz)
ds = pcr.Dataset.from_ratings(R)
idx, item, val = ds.csr(0)
lens = np.diff(idx)
user = np.repeat(np.arange(R.d1), lens)
lev = np.rint(val).astype(np.int64)
s = pcr.Solver(ds, pcr.Parameter(k=a.k, precision=pcr.PCR_F32, do_predict=0, **{"lambda": 5000.0}))
s.set_factors(pcr.initial(R.d1, a.k), pcr.initial(R.d2, a.k))


def order(m):
    """positions sorted by (user, level, m): the (level, m) order of every user, concatenated"""
    return np.lexsort((m, lev, user))


def report(tag, prev_perm, m_new):
    mo, lo, uo = m_new[prev_perm], lev[prev_perm], user[prev_perm]
    same = (uo[1:] == uo[:-1]) & (lo[1:] == lo[:-1])
    inv = same & (mo[1:] < mo[:-1])
    inv_per_user = np.bincount(uo[1:][inv], minlength=R.d1)
    sorted_users = inv_per_user == 0
    new_perm = order(m_new)
    rank_old = np.empty(R.nnz, np.int64); rank_old[prev_perm] = np.arange(R.nnz)
    rank_new = np.empty(R.nnz, np.int64); rank_new[new_perm] = np.arange(R.nnz)
    disp = np.abs(rank_new - rank_old)
    dmax_user = np.zeros(R.d1, np.int64); np.maximum.at(dmax_user, user, disp)
    w = lens / lens.sum()
    print(f"{tag:14s} users still sorted {100 * sorted_users.mean():5.1f} % (rating-weighted {100 * (w * sorted_users).sum():5.1f} %)  "
          f"inverted adjacent pairs {100 * inv.sum() / max(same.sum(), 1):5.2f} %  displacement: median of user maxima {int(np.median(dmax_user)):4d}, "
          f"rating-weighted mean of (max displacement / length) {(w * dmax_user / np.maximum(lens, 1)).sum():.3f}, "
          f"users with max displacement <= 8: {100 * (dmax_user <= 8).mean():5.1f} % (weighted {100 * (w * (dmax_user <= 8)).sum():5.1f} %)", flush=True)
    return new_perm


perm = order(s.comp_m())
for it in range(1, a.t + 1):
    s.update_V()
    perm = report(f"iter {it:2d} V step", perm, s.comp_m())
    s.update_U()
    perm = report(f"iter {it:2d} U step", perm, s.comp_m())

#!/usr/bin/env python3
"""Dev tool (GPU box): the two lower bounds of the U step's fork..join time on the ml1m bench workload (NOTES round 4).

  chain   per length class, the Newton step of its LONGEST user with the chip to itself (a data set of that one user, same V):
          the serial chain of row passes x rounds of dependent L2 round trips + sorts that no placement can shorten
  alone   the whole class with the chip to itself (a data set of its users only, same V): chain + its own queueing
  work    rows gathered by all classes x row bytes / the L2 gather ceiling (and / the rate the dedicated gather kernels reach)
  all     the U step as shipped (every class side by side): fork..join wall and the per-class times in company

The factors are those of iteration `--at` of the bench trajectory (so CG / line-search counts are the bench's); every subset
solver takes U (its users' rows) and V from that state, runs comp_m (the sorted state) and ONE timed update_U.
usage: exp_ustep_bound.py [--at 8] [--tune key=value ...]"""
import argparse, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import primalcr_amd as pcr
from primalcr_amd import synth

ap = argparse.ArgumentParser()
ap.add_argument("--at", type=int, default=8)
ap.add_argument("--tune", action="append", default=[])
ap.add_argument("--reps", type=int, default=5)
args = ap.parse_args()
tune = dict(kv.split("=", 1) for kv in args.tune)
r, lam = 100, 5000.0
R = synth.generate("ml1m")
ds = pcr.Dataset.from_ratings(R)
idx, item, val = ds.csr(0)
lens = np.diff(idx)
with pcr.tuned(**dict(tune, count_rows=1)):
    full = pcr.Solver(ds, pcr.Parameter(k=r, do_predict=0, **{"lambda": lam}))
full.set_factors(pcr.initial(R.d1, r), pcr.initial(R.d2, r))
full.iterate(args.at)
U, V = full.get_factors()
rows0 = full.counter("ustep_row_gathers")
full.profile(True, period=1); full.profile_reset()
full.iterate(args.reps)
p = full.profile_all()
avg = lambda pr, k: 1e3 * pr[k][0] / max(pr[k][1], 1) if k in pr else float("nan")
rows = (full.counter("ustep_row_gathers") - rows0) / args.reps
classes = full.ustep_classes()
print(f"all classes side by side (iterations {args.at + 1}..{args.at + args.reps}): fork..join {avg(p, 'wall:ustep'):.1f} us; rows gathered per U step "
      f"{rows / 1e6:.2f} M = {rows * r * 4 / 1e9:.2f} GB -> {rows * r * 4 / 18.8e12 * 1e6:.0f} us at the 18.8 TB/s L2 gather ceiling, "
      f"{rows * r * 4 / 14.5e12 * 1e6:.0f} us at the 14.5 TB/s of k_sddmm / k_spmm")
for c in classes:
    print(f"   {c:18s} in company {avg(p, c):7.1f} us   users {full.profile_scope(c)[1]:5d}  ratings {full.profile_scope(c)[0]:7d}")
full.close()


def subset(users, knobs):
    """U step of a data set holding only `users` (original ids), factors of the bench trajectory: (wall us, {class: us})."""
    users = np.sort(np.asarray(users))
    mask = np.zeros(R.d1, bool); mask[users] = True
    keep = np.repeat(mask, lens)
    nidx = np.concatenate([[0], np.cumsum(lens[mask])]).astype(np.int64)
    d = pcr.Dataset.from_csr(len(users), R.d2, nidx, item[keep].astype(np.int32), val[keep])
    with pcr.tuned(**dict(tune, **knobs)):
        s = pcr.Solver(d, pcr.Parameter(k=r, do_predict=0, **{"lambda": lam}))
    walls, per = [], {}
    for rep in range(args.reps + 1):
        s.set_factors(U[users], V)
        s.comp_m(want=False)
        if rep == 1:
            s.profile(True, period=1); s.profile_reset()
        s.update_U()
    pr = s.profile_all()
    out = (avg(pr, "wall:ustep"), {c: avg(pr, c) for c in s.ustep_classes()})
    s.close()
    return out


order = np.argsort(-lens, kind="stable")
long_users = order[lens[order] > 1024]
groups = [("cluster (16 longest, K = 4)", long_users[:16], {}),
          ("1025..4096 without them", long_users[16:], {"cluster_k": 1, "ustep_mode": 1}),
          ("513..1024", np.where((lens > 512) & (lens <= 1024))[0], {"cluster_k": 1}),
          ("129..512", np.where((lens > 128) & (lens <= 512))[0], {}),
          ("65..128", np.where((lens > 64) & (lens <= 128))[0], {}),
          ("33..64", np.where((lens > 32) & (lens <= 64))[0], {}),
          ("<= 32", np.where(lens <= 32)[0], {})]
print(f"\n{'class':30s} {'users':>6s} {'longest':>8s} {'chain: longest user alone':>28s} {'class alone':>14s}")
for name, users, knobs in groups:
    if len(users) == 0:
        continue
    longest = users[np.argmax(lens[users])]
    w1, c1 = subset([longest], knobs)
    wa, ca = subset(users, knobs)
    print(f"{name:30s} {len(users):6d} {int(lens[longest]):8d} {w1:20.1f} us ({','.join(c1)}) {wa:10.1f} us ({','.join(ca)})")

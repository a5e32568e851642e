#!/usr/bin/env python3
"""Loader scaling (SURVEY 8f-2, round-4 verdict item 3): write an N-rating Netflix-shaped text directory with the C++ writer, time
pcr_dataset_load_mt at several thread counts (best of `--repeat`), check the CSR against the generator's.

    python tools/exp_loader.py --nnz 10000000 --threads 1,2,4,8,16 [--shuffle] [--dir /tmp/pcr_loader]

--shuffle: the training file's lines in random order (the general, unsorted path of the CSR build)."""
import argparse, os, sys, time, shutil
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import primalcr_amd as pcr
from primalcr_amd import synth

ap = argparse.ArgumentParser()
ap.add_argument("--nnz", type=int, default=10_000_000)
ap.add_argument("--threads", default="1,2,4,8,16")
ap.add_argument("--repeat", type=int, default=3)
ap.add_argument("--shuffle", action="store_true")
ap.add_argument("--dir", default="/tmp/pcr_loader")
ap.add_argument("--keep", action="store_true")
a = ap.parse_args()
s = synth.SHAPES["netflix"]
d1 = max(64, int(s[0] * a.nnz / s[2]))
t0 = time.time()
R = synth.generate_fast("netflix", d1=d1, nnz=a.nnz)
t1 = time.time()
synth.write_dir(R, a.dir)
if a.shuffle:
    p = os.path.join(a.dir, "training.ratings")
    lines = open(p, "rb").read().split(b"\n")[:-1]
    rng = np.random.default_rng(1)
    order = rng.permutation(len(lines))
    open(p, "wb").write(b"\n".join(lines[i] for i in order) + b"\n")
sz = os.path.getsize(os.path.join(a.dir, "training.ratings"))
print(f"[data] {d1} users x {R.d2} items, {R.nnz} ratings, {sz / 1e6:.1f} MB of text (generated {t1 - t0:.1f}s, written {time.time() - t1:.1f}s)")
ref = None
for th in [int(x) for x in a.threads.split(",")]:
    best = 1e9
    for _ in range(a.repeat):
        t = time.perf_counter()
        ds = pcr.Dataset.load(a.dir, threads=th)
        best = min(best, time.perf_counter() - t)
    idx, item, val = ds.csr(0)
    ok = np.array_equal(idx, R.index) and np.array_equal(item, R.item) and np.array_equal(val, R.val)
    tidx, titem, tval = ds.csr(1)
    ok = ok and np.array_equal(tidx, R.tindex) and np.array_equal(titem, R.titem) and np.array_equal(tval, R.tval)
    print(f"threads {th:3d}: {best:7.3f} s  {R.nnz / best / 1e6:8.1f} M ratings/s  {sz / best / 1e6:8.1f} MB/s  csr {'ok' if ok else 'MISMATCH'}")
if not a.keep:
    shutil.rmtree(a.dir, ignore_errors=True)

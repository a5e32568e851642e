#!/usr/bin/env python3
"""Generate tests/golden/ml1m_test.{npz,json}: the reference's own shipped rating file ml1m/test.ratings (the only real
MovieLens data in the checkout: 60 400 ratings, 10 per user) used as training AND test set, and what the UNMODIFIED
reference binary (oracle/_ref/omp-pmf-train -n 1, deterministic) prints for it -- the known-answer runs of BASELINE.md
section 2.  TEST INFRASTRUCTURE; run in the build container where /root/reference exists.  Only data is stored."""
import json, os, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.oracle_py import REF_TRAIN  # noqa: E402
SRC = "/root/reference/ml1m/test.ratings"
u, i, v = np.loadtxt(SRC, dtype=np.int64, unpack=True)
assert u.shape[0] == 60400 and np.all(np.diff(u) >= 0)
runs = {}
with tempfile.TemporaryDirectory() as td:
    d = os.path.join(td, "data"); os.makedirs(d)
    for name in ("training.ratings", "test.ratings"):
        with open(os.path.join(d, name), "w") as f:
            f.write(open(SRC).read())
    open(os.path.join(d, "meta"), "w").write("6040 3952\n60400 training.ratings\n60400 test.ratings\n")
    for tag, args in (("s2_l5000", ["-s", "2", "-l", "5000"]), ("s2_l50", ["-s", "2", "-l", "50"]), ("s1_l50", ["-s", "1", "-l", "50"])):
        out = subprocess.run([REF_TRAIN, *args, "-k", "10", "-n", "1", "-t", "3", "-p", "1", d, os.path.join(td, "m.model")],
                             cwd=td, capture_output=True, text=True, check=True).stdout
        runs[tag] = out
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "ml1m_test.npz"), user=(u - 1).astype(np.int16), item=(i - 1).astype(np.int16), val=v.astype(np.int8))
json.dump({"d1": 6040, "d2": 3952, "k": 10, "iters": 3, "stdout": runs}, open(os.path.join(ROOT, "tests", "golden", "ml1m_test.json"), "w"), indent=1)
for k, o in runs.items():
    print(k); print("\n".join(l for l in o.split("\n") if l.startswith(("Iter", "(T"))))

#!/usr/bin/env python3
"""Generate tests/golden/toy_test.{npz,json}: the reference's shipped toy-example/test.ratings (100 020 REAL-VALUED ratings
in about +-4, i.e. 9 lround levels, non-positive gains in the NDCG; its first 1500 users) used as training AND test set -- configs[0] of
BASELINE.json (toy-example, PrimalCR++ -k 10 -n 1) -- and what the UNMODIFIED reference binary prints for it.
TEST INFRASTRUCTURE; run in the build container where /root/reference exists.  Only data is stored."""
import json, os, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.oracle_py import REF_TRAIN  # noqa: E402
SRC = "/root/reference/toy-example/test.ratings"
u, i, v = [], [], []
for line in open(SRC):
    a, b, c = line.split()
    u.append(int(a)); i.append(int(b)); v.append(float(c))
u = np.array(u); i = np.array(i); v = np.array(v, np.float64)
assert np.all(np.diff(u) >= 0)
NU = 1500                                    # the first 1500 users (24 862 ratings) keep the fixture small
keep = u <= NU
u, i, v = u[keep], i[keep], v[keep]
lines = open(SRC).read().split("\n")[:len(u)]
runs = {}
with tempfile.TemporaryDirectory() as td:
    d = os.path.join(td, "data"); os.makedirs(d)
    for name in ("training.ratings", "test.ratings"):
        open(os.path.join(d, name), "w").write("\n".join(lines) + "\n")
    open(os.path.join(d, "meta"), "w").write(f"{NU} 3952\n{len(u)} training.ratings\n{len(u)} test.ratings\n")
    for tag, args in (("s2", ["-s", "2"]), ("s1", ["-s", "1"])):
        runs[tag] = subprocess.run([REF_TRAIN, *args, "-k", "10", "-n", "1", "-t", "2", "-p", "1", d, os.path.join(td, "m.model")],
                                   cwd=td, capture_output=True, text=True, check=True).stdout
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "toy_test.npz"), user=(u - 1).astype(np.int16), item=(i - 1).astype(np.int16), val=v)
json.dump({"d1": NU, "d2": 3952, "k": 10, "iters": 2, "lam": 5000.0, "stdout": runs}, open(os.path.join(ROOT, "tests", "golden", "toy_test.json"), "w"), indent=1)
for k, o in runs.items():
    print(k); print("\n".join(l for l in o.split("\n") if l.startswith(("Iter", "(T"))))

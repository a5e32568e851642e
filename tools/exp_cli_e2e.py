#!/usr/bin/env python3
"""The drop-in CLI end to end at size (round-4 verdict item 2): write the shape's text directory (meta + rating files, the
reference's format) with the C++ generator / writer, run primalcr_amd/bin/omp-pmf-train -k K -l 5000 -t 10 (defaults: -p 1) with
--timing, then the unmodified reference binary on the same directory -- bounded on large shapes (--ref-iters 1 --ref-predict 0 for
the Netflix shape: one iteration, no evaluation; said in the record).

    python tools/exp_cli_e2e.py --shape netflix --out profiles/r05_cli_netflix.json [--ref-iters 1 --ref-predict 0] [--no-reference]"""
import argparse, json, os, shutil, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from primalcr_amd import synth

ap = argparse.ArgumentParser()
ap.add_argument("--shape", default="netflix")
ap.add_argument("--users", type=int, default=None)
ap.add_argument("--nnz", type=int, default=None)
ap.add_argument("--rank-k", type=int, default=100)
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--ref-iters", type=int, default=1)
ap.add_argument("--ref-predict", type=int, default=0)
ap.add_argument("--no-reference", action="store_true")
ap.add_argument("--extra", default="", help="extra options for our CLI, space-separated")
ap.add_argument("--dir", default="/tmp/pcr_cli_data")
ap.add_argument("--out", default=None)
a = ap.parse_args()
t0 = time.time()
R = synth.generate_fast(a.shape, d1=a.users, nnz=a.nnz)
t1 = time.time()
synth.write_dir(R, a.dir)
t2 = time.time()
size = sum(os.path.getsize(os.path.join(a.dir, f)) for f in os.listdir(a.dir))
print(f"[data] {a.shape}: {R.d1} x {R.d2}, {R.nnz} ratings, {size / 1e6:.0f} MB of text (generated {t1 - t0:.1f}s, written {t2 - t1:.1f}s)", file=sys.stderr, flush=True)
rec = bench.cli_leg(a.dir, a.rank_k, 5000.0, iters=a.iters, ref_iters=a.ref_iters, ref_predict=a.ref_predict, extra=tuple(a.extra.split()),
                    run_reference=not a.no_reference, timeout_s=3000)
rec["workload"] = f"{a.shape}-shaped: {R.d1} users x {R.d2} items, {R.nnz} ratings, {len(R.tval)} test ratings, k = {a.rank_k}; text directory {size} bytes"
rec["host_cores"] = bench.host_cores()
shutil.rmtree(a.dir, ignore_errors=True)
txt = json.dumps(rec, indent=1)
print(txt)
if a.out:
    open(a.out, "w").write(txt + "\n")

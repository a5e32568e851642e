#!/usr/bin/env python3
"""bench.py -- the headline measurement (BASELINE.json): PrimalCR++ pairwise-comparisons/sec
+ NDCG@10 on ml1m-shaped synthetic ratings, rank 100, lambda 5000, on N MI355X.

    python bench.py --gpus N --steps K --warmup W

N > 1 without a launcher: the N ranks are started as direct CHILD processes (RANK / WORLD_SIZE / MASTER_* in
their environment, 127.0.0.1 rendezvous) before this process imports torch or touches a GPU, and rank 0's
JSON line is relayed; started under torch.distributed.run it is one of the ranks.  A rank that stops -- exit
code, signal, exception -- is named on stderr, takes the job down, and the last stdout line is {"error": ...}.

A "step" is one outer iteration of pcrpp() (pcrpp.cpp:873-881): one truncated-Newton step on V
(gradient, <=10 CG Hessian-vector products, line search) and one Newton step per user on U -- the
same clock scope as the reference's "Iter k time" (no load, no init, no evaluation).  Inputs
(ratings, factors) are resident in HBM before the timed region starts.

value = #Omega * K / seconds, #Omega = #{(i,j,k): R_ij > R_ik} = the ordered pairs the objective
sums over (all ranks).  User-sharded (pcrpp.cpp:825-833): every rank generates and holds ONLY its own users
(pcr_solver_create_shard), V and the CG vectors are replicated, the V-gradient and every Hessian-vector
product are all-reduced over RCCL.
  --shape ml1m     WEAK scaling: rank q owns 6040 users / 939 809 ratings of its own (the numpy generator's set of seed
                   SEED + q over the item catalogue of seed SEED; rank 0's block is the N = 1 workload, configs[1])
  --shape netflix  configs[3]: 480 189 x 17 770, 100 M ratings, nnz-balanced user ranges (strong scaling)
  --shape yahoo    configs[4]: 1.8 M x 136 k, 700 M ratings, k = 200: the first 225 000 N users (N = 8: all of them)

The JSON line also carries
  roofline      the kernel slot with the most GPU time: algorithmic bytes per launch / its average duration
                from HIP events on its launch stream, against the 8 TB/s HBM3E peak -- and `binding`: the level of the
                memory hierarchy that actually serves its row gathers (l2-gather / mall-gather / hbm-gather by the size of
                the gathered table), its measured ceiling from MI355X_MICROARCH.md and the fraction of THAT
  cpu_baseline  the reference's own OpenMP path (oracle/_ref/omp-pmf-train, built from the unmodified reference) on the same
                data on this host's cores (rank 0, N = 1 only); the single-thread C restatement when _ref is absent
  f64           the same K steps with fp64 storage (the reference's arithmetic type), with its own roofline blocks
  netflix       (default N = 1 run) configs[3] on this GPU: a few steps of the Netflix-shaped set, same fields
"""
import argparse
import json
import os
import re
import socket
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)
# MI355X_MICROARCH.md, "Indexed rows": measured chip-wide row-gather ceilings by where the table is served from
GATHER_CEILING_GBS = {"l2-gather": 18800.0,     # table shared by every workgroup, in each XCD's 4 MiB L2: 16.8-18.8 TB/s
                      "mall-gather": 8600.0,    # 38 MB table, uniformly random rows (Infinity Cache): 8.6 TB/s; 151 MB: 7.4-7.9
                      "hbm-gather": 6100.0}     # tables beyond the 256 MiB Infinity Cache: 6.0-6.1 TB/s
TRAFFIC_FILE = "r06_traffic.json"   # stored PMC passes of this command (tools/collect_profiles.sh): the fall-back when the live passes are off
USERS_PER_GPU = 6040
D2, NNZ_PER_GPU = 3952, 939809
YAHOO_USERS_PER_GPU = 225000


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def gather_level(table_bytes):
    """Which level serves uniformly random row gathers from a table of that size (MI355X_MICROARCH.md: 4 MiB L2 per XCD,
    every XCD caching its own copy; 256 MiB Infinity Cache)."""
    if table_bytes <= (4 << 20):
        return "l2-gather"
    if table_bytes <= (256 << 20):
        return "mall-gather"
    return "hbm-gather"


def gather_ceiling(table_bytes):
    """(level, ceiling GB/s, L2 share) for uniformly random row gathers from a table of that size.  MI355X_MICROARCH.md,
    Indexed rows: "an XCD's 4 MiB L2 holds 4 MiB / T of a uniformly gathered table of T bytes; the time scales with the reads
    that go beyond L2" -- so a table a little larger than one L2 (the Netflix shape's 7.1 MB) is served partly at the L2 rate
    and partly at the next level's: ceiling = 1 / (h / L2 rate + (1 - h) / next level's rate), h = min(1, 4 MiB / T)."""
    lv = gather_level(table_bytes)
    h = min(1.0, (4 << 20) / max(table_bytes, 1))
    if lv == "l2-gather":
        return lv, GATHER_CEILING_GBS[lv], 1.0
    c = 1.0 / (h / GATHER_CEILING_GBS["l2-gather"] + (1.0 - h) / GATHER_CEILING_GBS[lv])
    return lv, round(c, 1), round(h, 3)


def algorithmic_bytes(slot, nnz_b, nu_b, d2, r, esz):
    """Compulsory HBM bytes of ONE launch of a per-user kernel over a length bin holding nnz_b
    ratings of nu_b users (ideal caching: every operand crosses HBM once).  DESIGN.md section 3.5."""
    F_U = nu_b * r * esz           # the bin's user factors
    F_V = d2 * r * esz             # one item-side matrix
    cls = slot.split("/")[0]
    if cls == "sddmm":             # k_sddmm: read item ids, user ids, U, one item-side matrix; write one score per rating
        return nnz_b * (4 + 4 + esz) + F_U + F_V
    if cls == "prepare":           # k_prepare: read m,lvl,item; write ms,sitem,slvl,sidx; objp  (the window cache is extra)
        return nnz_b * (esz + 2 + 4) + nnz_b * (esz + 4 + 2 + 4) + nu_b * 24
    if cls == "vgrad":             # k_vsweep<GRAD>: read ms,sidx; write c
        return nnz_b * (esz + 4 + esz) + nu_b * 16
    if cls == "vhv":               # k_vsweep<HV>: read b,sidx; write c
        return nnz_b * (esz + 4 + esz) + nu_b * 16
    if cls == "spmm":              # k_spmm: read c (through the static CSC->CSR map), user|flag word, U rows; write one row per item
        return nnz_b * (esz + 4 + 4) + F_U + F_V
    if cls == "spmm_fin":          # k_spmm_fin: read slab + base, write out
        return 3 * F_V
    if cls == "ustep":             # k_ustep: read ms,sitem,slvl,U,V; write U,objp and the sorted state of u_new (ms,sitem,slvl,sidx)
        return nnz_b * (esz + 4 + 2) + nnz_b * (esz + 4 + 2 + 4) + nu_b * 24 + 2 * F_U + F_V
    if cls == "cg":                # k_cg_bc: read p,Hp,rr,delta, write delta,rr,p
        return 7 * F_V
    return 0


# HIP-event slot -> kernel symbol prefix in a rocprofv3 trace (profiles/, tools/pmc_traffic.py)
SLOT_KERNEL = {"sddmm": "void k_sddmm<", "spmm": "void k_spmm<", "spmm_fin": "void k_spmm_fin<", "prepare": "void k_prepare",
               "vgrad": "void k_vsweep", "vhv": "void k_vsweep", "ustep": "void k_ustep<", "cg": "void k_cg_"}


def slot_kernel_match(slot, kernel_name, prec):
    """Does a rocprof kernel name belong to this HIP-event slot (class/workgroup size[.bound][g][c][#symbol id])?"""
    cls, _, tag = slot.partition("/")
    if not kernel_name.startswith(SLOT_KERNEL.get(cls, "\0")):
        return False
    args = kernel_name[kernel_name.index("<") + 1:kernel_name.index(">")].replace(" ", "").split(",")
    if args[0] != ("float" if prec == "f32" else "double"):
        return False
    if kernel_name.startswith("void k_prepare_all<"):          # both LDS classes in one launch: slot "prepare/all"
        return cls == "prepare" and tag == "all"
    if cls == "prepare" and tag == "all":
        return False
    if kernel_name.startswith("void k_vsweep_all<"):           # both LDS classes in one launch: slot tag "all"
        return cls in ("vgrad", "vhv") and tag == "all" and (args[1] == "true") == (cls == "vhv")
    if kernel_name.startswith("void k_vsweep_wave<"):          # one wave per user: slot tag "64"
        return cls in ("vgrad", "vhv") and tag == "64" and (args[1] == "true") == (cls == "vhv")
    if cls in ("vgrad", "vhv") and not kernel_name.startswith("void k_vsweep<"):
        return False
    if not tag:
        return True
    tag, _, sym = tag.partition("#")                           # k_ustep: the symbol id of classes that share a workgroup form
    flags = tag.lstrip("0123456789.")
    block = tag[:len(tag) - len(flags)].partition(".")[0]
    big, clu = "g" in flags, "c" in flags
    if args[1] != block or (args[2] == "true") != big:
        return False
    if cls in ("vgrad", "vhv"):
        return (args[3] == "true") == (cls == "vhv")
    if cls == "ustep":
        if clu or args[3] != "1":
            return clu and args[3] != "1"
        # l = latency form (8 rows in flight), r = one-wave class with LDS-resident rows, #n = symbol id: one symbol per class
        return (args[5] == "8") == ("l" in flags) and (block != "64" or (args[4] == "true") == ("r" in flags)) and \
               (len(args) < 7 or args[6] == (sym or "0"))
    return True


def host_cores():
    """CPU cores this process may actually use: the cgroup quota when there is one (a GPU box hands
    each GPU a 16-core share of a 256-thread host), else the affinity mask."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(round(int(quota) / int(period)))))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(R, n_pairs, r, lam, sample_ratings=2_000_000, single_thread=True, data_dir=None):
    """Reference OpenMP path on this host (kind "reference"), else the C restatement (kind "port").
    BASELINE.md 3.3: timed at -n <all cores of this box's share> AND at -n 1.
    Bounded: data sets beyond `sample_ratings` ratings are timed on a prefix of their users (same shape, fewer users)."""
    from oracle import oracle_py
    from primalcr_amd import synth
    cores = host_cores()
    sample_note = "the full data set"
    user, tuser = R.user, R.tuser
    if R.nnz > sample_ratings:
        nu = int(user[sample_ratings])            # whole users within the first `sample_ratings` ratings (triplets are user-sorted)
        keep, tkeep = user < nu, tuser < nu
        R = synth.Ratings(nu, R.d2, user[keep], R.item[keep], R.val[keep], tuser[tkeep], R.titem[tkeep], R.tval[tkeep])
        n_pairs = synth.count_pairs(R)
        sample_note = f"its first {nu} users ({R.nnz} ratings, {n_pairs} ordered pairs)"
    elif not isinstance(R, synth.Ratings):
        R = synth.Ratings(R.d1, R.d2, user, R.item, R.val, tuser, R.titem, R.tval)
    if os.path.exists(oracle_py.REF_TRAIN):
        def ref_run(threads, iters, d, td):
            t0 = time.time()
            out = subprocess.run([oracle_py.REF_TRAIN, "-s", "2", "-k", str(r), "-l", repr(lam), "-t", str(iters),
                                  "-p", "0", "-n", str(threads), d, os.path.join(td, "m.model")],
                                 cwd=td, capture_output=True, text=True, check=True).stdout
            times = [float(x) for x in re.findall(r"^Iter \d+ time (\S+) obj", out, re.M)]
            log(f"[cpu_baseline] reference omp-pmf-train -n {threads}: {times[-1]:.2f}s for {iters} iteration(s) (wall {time.time() - t0:.1f}s)")
            return times[-1]
        with tempfile.TemporaryDirectory() as td:
            # (data_dir: the text directory of the WHOLE set, already written by the caller -- used when nothing was cut)
            d = data_dir if (data_dir and sample_note == "the full data set") else synth.write_dir(R, os.path.join(td, "data"))
            it_all, it_one = 2, 1
            secs = ref_run(cores, it_all, d, td)
            secs1 = ref_run(1, it_one, d, td) if single_thread else None
        out = {"value": n_pairs * it_all / secs, "unit": "pairs/s", "cores": cores, "kind": "reference",
               "sample": f"omp-pmf-train -s 2 -k {r} -l {lam:g} -t {it_all} -p 0 -n {cores} on {sample_note}; "
                         f"'Iter {it_all} time' = {secs:.3f} s", "s_per_iter": secs / it_all}
        if secs1 is not None:
            out["single_thread"] = {"value": n_pairs * it_one / secs1, "unit": "pairs/s", "cores": 1, "s_per_iter": secs1 / it_one,
                                    "sample": f"the same command with -n 1 -t {it_one}: 'Iter {it_one} time' = {secs1:.3f} s"}
        return out
    orc = oracle_py.Oracle()
    nu = 400
    keep = R.user < nu
    X = orc.build_csr(nu, R.d2, R.user[keep], R.item[keep], R.val[keep])
    U, V = orc.initial(nu, r), orc.initial(R.d2, r)
    _, _, recs = orc.train(X, U, V, lam, 1, do_predict=0)
    secs = recs[1]["seconds"]
    pairs = orc.count_pairs(X)
    return {"value": pairs / secs, "unit": "pairs/s", "cores": 1, "kind": "port",
            "sample": f"C restatement, first {nu} users ({int(keep.sum())} ratings, {pairs} pairs), 1 iteration = {secs:.2f} s"}


def cli_leg(data_dir, r, lam, iters=10, ref_iters=None, ref_predict=1, threads=None, extra=(), timeout_s=1500, run_reference=True):
    """The drop-in CLI end to end (the reference's user-facing unit, pmf-train.cpp:247-314): `omp-pmf-train -k r -l lam -t iters
    data_dir model` with the reference's defaults otherwise (-p 1: both evaluations after every iteration), as a child process in
    a scratch directory, wall-clocked from before the fork to after the exit, with the product's own phase split (--timing: load /
    init / create / train / iter / eval / write); then the UNMODIFIED reference binary (oracle/_ref/omp-pmf-train -n <cores>) on the
    same directory, same clock.  ref_iters / ref_predict bound the reference's run on large shapes (stated in the record)."""
    import shutil
    from oracle import oracle_py
    ours = os.path.join(ROOT, "primalcr_amd", "bin", "omp-pmf-train")
    cores = threads or host_cores()
    out = {"command": f"omp-pmf-train -k {r} -l {lam:g} -t {iters} -n {cores} --timing <dir> <model>  (defaults otherwise: -s 2 -p 1)"}

    def last_metrics(text):
        te = re.findall(r"^\(Testing\) pairwise error is (\S+) and ndcg is (\S+)", text, re.M)
        ob = re.findall(r"^Iter (\d+) time (\S+) obj (\S+)", text, re.M)
        return {"ndcg10_test": float(te[-1][1]) if te else None, "pairwise_error_test": float(te[-1][0]) if te else None,
                "iterations": int(ob[-1][0]) if ob else None, "iter_time_s": float(ob[-1][1]) if ob else None,
                "objective": float(ob[-1][2]) if ob else None}
    td = tempfile.mkdtemp(prefix="pcr_cli_", dir="/tmp")
    try:
        t0 = time.perf_counter()
        p = subprocess.run([ours, "-k", str(r), "-l", repr(lam), "-t", str(iters), "-n", str(cores), "--timing", *extra, data_dir,
                            os.path.join(td, "ours.model")], cwd=td, capture_output=True, text=True, timeout=timeout_s)
        out["wall_s"] = time.perf_counter() - t0
        if p.returncode != 0:
            out["error"] = f"omp-pmf-train exited with {p.returncode}: {p.stderr.strip()[-300:]}"
            return out
        m = re.search(r"^\[timing\] (.*)$", p.stderr, re.M)
        ph = {k: float(v) for k, v in (kv.split("=") for kv in m.group(1).split())} if m else {}
        out.update({k: ph.get(k) for k in ("load_s", "init_s", "create_s", "train_s", "iter_s", "eval_s", "write_s")})
        mc = re.search(r"^\[timing-create\] (.*)$", p.stderr, re.M)       # where create_s went: runtime wait, the library's set-up phases, factor upload
        if mc:
            out["create_split"] = {k: float(v) for k, v in (kv.split("=") for kv in mc.group(1).split())}
        out["process_s"] = ph.get("wall_s")              # main() start to end; wall_s - process_s = fork / exec / library + code-object load / exit
        out["model_bytes"] = os.path.getsize(os.path.join(td, "ours.model"))
        out.update(last_metrics(p.stdout))
        if ph:
            big = max(("load_s", "init_s", "create_s", "iter_s", "eval_s", "write_s"), key=lambda k: ph.get(k, 0.0))
            out["largest_phase"] = big
        if run_reference and os.path.exists(oracle_py.REF_TRAIN):
            ri = ref_iters or iters
            t0 = time.perf_counter()
            q = subprocess.run([oracle_py.REF_TRAIN, "-k", str(r), "-l", repr(lam), "-t", str(ri), "-p", str(ref_predict), "-n", str(cores), data_dir,
                                os.path.join(td, "ref.model")], cwd=td, capture_output=True, text=True, timeout=timeout_s)
            rw = time.perf_counter() - t0
            if q.returncode == 0:
                rm = last_metrics(q.stdout)
                out["reference"] = dict(rm, wall_s=rw, cores=cores,
                                        command=f"oracle/_ref/omp-pmf-train -k {r} -l {lam:g} -t {ri} -p {ref_predict} -n {cores} (the unmodified reference)")
                out["reference_wall_s"] = rw
                if ri == iters and ref_predict == 1:
                    out["speedup_wall"] = rw / out["wall_s"]
                    if rm["ndcg10_test"] is not None and out.get("ndcg10_test") is not None:
                        out["ndcg10_test_minus_reference"] = out["ndcg10_test"] - rm["ndcg10_test"]
            else:
                out["reference"] = {"error": f"exited with {q.returncode}: {q.stderr.strip()[-200:]}"}
    finally:
        shutil.rmtree(td, ignore_errors=True)
    return out


def pmc_accumulate(csv_path, ctr, acc):
    """One rocprofv3 --pmc pass (its *counter_collection.csv: one row per dispatch, counter and -- on multi-XCD parts -- instance)
    summed per kernel into acc[kernel][counter]; acc[kernel]["launches_<ctr>"] counts the dispatches of this pass."""
    import csv
    seen = set()
    for row in csv.DictReader(open(csv_path)):
        k = row["Kernel_Name"]
        acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
        d = (row.get("Dispatch_Id"), k)
        if d not in seen:
            seen.add(d)
            acc[k]["launches_" + ctr] += 1


PMC_READ, PMC_WRITE = "TCC_EA0_RDREQ_DRAM_32B_sum", "TCC_EA0_WRREQ_WRITE_DRAM_32B_sum"
# kernels of the training iteration (set-up, evaluation and probe kernels are not part of a step's traffic)
HOT_KERNEL = re.compile(r"^(void )?k_(sddmm|spmm|prepare|vsweep|ustep|cg_|axpy|obj|fin|dots|sum_stage|unewton|vblock|p2p)")


def pmc_per_launch(acc):
    """{kernel: launches, read and written bytes per launch} from one pass of the two 32-byte-unit counters (profiles/
    r06_dram_calib.md: TCC_EA0_RDREQ_DRAM_32B_sum x 32 B and TCC_EA0_WRREQ_WRITE_DRAM_32B_sum x 32 B are byte-exact on streamed
    reads, writes and whole-row gathers -- 128-byte read requests count 4, 64-byte writes 2 -- and need no gfx950 correction,
    unlike FETCH_SIZE, which tallies a 128-byte request at 64).  They are the L2s' requests to local memory: Infinity-Cache hits
    included, like every memory-side TCC counter of this part."""
    kernels = {}
    for k, v in acc.items():
        n = v.get("launches_" + PMC_READ, 0)
        if n:
            kernels[k] = {"launches": int(n), "read_bytes_per_launch": 32.0 * v.get(PMC_READ, 0.0) / n,
                          "write_bytes_per_launch": 32.0 * v.get(PMC_WRITE, 0.0) / n}
    return kernels


def traffic_bytes(t):
    """Fabric-side bytes of one launch from a PMC record: the exact counters of round 6, or a stored FETCH_SIZE / WRITE_SIZE pass of
    earlier rounds (FETCH_SIZE doubled: it tallies gfx950's 128-byte read requests at 64 B, MI355X_MICROARCH.md)."""
    if "read_bytes_per_launch" in t:
        return int(t["read_bytes_per_launch"] + t["write_bytes_per_launch"])
    return int(2 * t["fetch_bytes_per_launch_raw"] + t["write_bytes_per_launch"])


def live_traffic(shape, prec, r, steps=10, warmup=5, timeout_s=170, users=None):
    """Memory-side traffic of THIS box, now: ONE rocprofv3 --pmc pass (both 32-byte-unit DRAM counters, no trace domain beside them,
    the program directly behind "--": MI355X_MICROARCH.md) of a short replay of the same workload in a child process (warmup +
    steps + 1 outer iterations from pcr_initial, no event timing).  Returns ({kernel name: {"launches", "read_bytes_per_launch",
    "write_bytes_per_launch"}}, iterations replayed, seconds) or (None, 0, reason)."""
    import collections, glob, shutil, signal
    exe = shutil.which("rocprofv3")
    if not exe:
        return None, 0, "rocprofv3 not found"
    t0 = time.time()
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    tmp = tempfile.mkdtemp(prefix="pcr_pmc_", dir="/tmp")
    try:
        out = os.path.join(tmp, "pmc")
        cmd = [exe, "--pmc", PMC_READ, PMC_WRITE, "--output-format", "csv", "-d", out, "--", sys.executable, os.path.abspath(__file__), "--shape", shape,
               "--precision", prec, "--rank-k", str(r), "--steps", str(steps), "--warmup", str(warmup), "--no-cpu", "--no-cli", "--no-f64", "--no-netflix", "--no-rows",
               "--no-profile", "--no-live-traffic", "--no-hbm", "--full-record", os.path.join(tmp, "child.json")] + (["--users", str(users)] if users else [])
        p = subprocess.Popen(cmd, cwd=tmp, env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL, stderr=subprocess.PIPE,
                             text=True, start_new_session=True)
        try:
            _, err = p.communicate(timeout=max(20.0, timeout_s))
        except subprocess.TimeoutExpired:
            os.killpg(p.pid, signal.SIGKILL)              # exactly the group this call started
            p.wait()
            return None, 0, "the counter pass did not finish within the time budget"
        if p.returncode != 0:
            return None, 0, f"the counter pass exited with {p.returncode}: {(err or '').strip()[-200:]}"
        files = glob.glob(os.path.join(out, "*", "*counter_collection.csv"))
        if not files:
            return None, 0, "the counter pass left no counter_collection.csv"
        pmc_accumulate(files[0], PMC_READ, acc)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    kernels = pmc_per_launch(acc)
    return (kernels, warmup + steps + (1 if warmup >= 1 else 0), time.time() - t0) if kernels else (None, 0, "no kernel appeared in the pass")


# ---- the HBM side (behind the Infinity Cache): the memory controllers' activity, sampled beside a sustained replay
# profiles/r06_umc_calib.md: /sys/class/drm/card*/device/mem_busy_percent (the SMU's average UMC activity) reads 75.4 % under a
# 6.25 TB/s streamed read of 1 GiB, 52.4 % under a 4.37 TB/s streamed write, 57.1 % under a 4.73 TB/s copy (82.9 / 83.5 / 82.7 GB/s
# per percent) and 0 % under 6-20 TB/s of reads that the Infinity Cache or the L2s serve (64 / 192 MiB re-read, row gathers from
# 1.6 / 7 / 109 MB tables): it is the one figure on this box that separates HBM from the Infinity Cache.
# The same activity as an accumulating counter (rocm_smi's "Memory Activity", percent x milliseconds): 76.3 / 53.2 / 63.3 % under
# 6.26 / 4.36 TB/s and the 1 GiB gather (82.0 GB/s per percent), 11 counts per second when idle or Infinity-Cache-resident -- a
# resolution of 0.001 % over a second where the sampled percentage has 1 %.  The sampler reads it before and after the replay.
HBM_GBS_PER_BUSY_PERCENT = 82.5


class HbmSampler:
    """mem_busy_percent of THIS process's GPU, sampled every few milliseconds by a thread while the main thread runs iterations."""

    def __init__(self, device):
        import ctypes, glob
        self.path, self.samples, self.thread, self.stop_flag = None, [], None, False
        try:
            hip = ctypes.CDLL("libamdhip64.so")
            buf = ctypes.create_string_buffer(64)
            if hip.hipDeviceGetPCIBusId(buf, 64, int(device)) == 0 and buf.value:
                bus = buf.value.decode().lower()
                for c in glob.glob("/sys/class/drm/card*/device/mem_busy_percent"):
                    if os.path.basename(os.path.realpath(os.path.dirname(c))).lower() == bus:
                        self.path = c
        except OSError:
            pass

        # rocm_smi's accumulating twin of that percentage, for the device with the same PCI address
        self.smi, self.smi_dev = None, None
        try:
            L = ctypes.CDLL("librocm_smi64.so")
            if self.path and L.rsmi_init(ctypes.c_uint64(0)) == 0:
                n = ctypes.c_uint32(0)
                L.rsmi_num_monitor_devices(ctypes.byref(n))
                want = os.path.basename(os.path.realpath(os.path.dirname(self.path))).lower()
                for dv in range(n.value):
                    bdf = ctypes.c_uint64(0)
                    if L.rsmi_dev_pci_id_get(ctypes.c_uint32(dv), ctypes.byref(bdf)) == 0:
                        b = bdf.value
                        if f"{(b >> 32) & 0xffffffff:04x}:{(b >> 8) & 0xff:02x}:{(b >> 3) & 0x1f:02x}.{b & 7:x}" == want:
                            self.smi, self.smi_dev = L, dv
        except OSError:
            pass

    def available(self):
        return self.path is not None

    def mem_activity_acc(self):
        """rocm_smi's accumulated memory-controller activity (percent x ms) of this GPU, or None."""
        if self.smi is None:
            return None
        import ctypes

        class Ctr(ctypes.Structure):
            _fields_ = [("type", ctypes.c_int), ("val", ctypes.c_uint64)]
        arr = (Ctr * 1)()
        arr[0].type = 1                                   # RSMI_COARSE_GRAIN_MEM_ACTIVITY
        ts = ctypes.c_uint64(0)
        if self.smi.rsmi_utilization_count_get(ctypes.c_uint32(self.smi_dev), arr, ctypes.c_uint32(1), ctypes.byref(ts)) != 0:
            return None
        return int(arr[0].val)

    def _loop(self, period):
        while not self.stop_flag:
            try:
                self.samples.append(int(open(self.path).read().strip()))
            except (OSError, ValueError):
                pass
            time.sleep(period)

    def start(self, period=0.02):
        # (only where the accumulating counter is not available: every read of the percentage is a query to the SMU, and 250 of
        # them per second slowed the replay itself -- 2.63 instead of 1.50 ms per ml1m step in the first run of this code)
        import threading
        self.samples, self.stop_flag, self.thread = [], False, None
        if self.smi is None:
            self.thread = threading.Thread(target=self._loop, args=(period,), daemon=True)
            self.thread.start()

    def stop(self):
        self.stop_flag = True
        if self.thread is None:
            return None, 0
        self.thread.join()
        xs = self.samples[len(self.samples) // 5:]        # (the firmware's average needs a moment to reach the replay's level)
        return (sum(xs) / len(xs), len(xs)) if xs else (None, 0)


class Job:
    """The process group (or none) and this rank's place in it."""

    def __init__(self, args, torch, dist, rank, N, device):
        self.args, self.torch, self.dist, self.rank, self.N, self.device = args, torch, dist, rank, N, device

    def allsum(self, x):
        if self.N == 1:
            return x
        t = self.torch.tensor([x], dtype=self.torch.float64, device="cuda" if self.args.rendezvous == "nccl" else "cpu")
        self.dist.all_reduce(t)
        return type(x)(t.item())

    def allmax(self, x):
        if self.N == 1:
            return x
        t = self.torch.tensor([x], dtype=self.torch.float64, device="cuda" if self.args.rendezvous == "nccl" else "cpu")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def bcast(self, obj):
        if self.N == 1:
            return obj
        box = [obj if self.rank == 0 else None]
        self.dist.broadcast_object_list(box, src=0)
        return box[0]


def make_shard(shape, job, users=None, nnz=None):
    """This rank's users of the workload, and nothing else: (ratings of the shard, first user, users of the job, notes)."""
    import primalcr_amd as pcr
    from primalcr_amd import synth
    rank, N = job.rank, job.N
    if shape == "ml1m":
        nu, nz = users or USERS_PER_GPU, nnz or NNZ_PER_GPU
        R = synth.generate("ml1m", seed=synth.SEED + rank, d1=nu, nnz=nz, item_seed=synth.SEED if rank else None)
        return R, rank * nu, nu * N, "weak", nz * N
    s = synth.SHAPES[shape]
    d1_shape = s[0] if (users is None or shape == "yahoo") else users       # netflix --users: a smaller shape; yahoo --users: a prefix
    ctr, _ = synth.generate_fast(shape, d1=None if shape == "yahoo" else d1_shape, nnz=nnz, counts_only=True)
    total = d1_shape if shape == "netflix" else min(s[0], users or YAHOO_USERS_PER_GPU * N)
    index = np.concatenate([[0], np.cumsum(ctr[:total])]).astype(np.int64)
    bounds = pcr.partition_users(index, N)                                   # nnz-balanced contiguous user ranges
    u0, u1 = int(bounds[rank]), int(bounds[rank + 1])
    R = synth.generate_fast(shape, d1=None if shape == "yahoo" else d1_shape, nnz=nnz, users=(u0, u1))
    return R, u0, total, "strong", int(index[-1])


def timed_run(job, ds, shard, d2, r, lam, prec, steps, warmup, profile, shm_name, count_rows=False, cold_start=True):
    """One solver, `warmup` untimed + exactly `steps` timed outer iterations (barrier + synchronize on both sides, MAX over
    ranks), then the quality after warmup + steps iterations."""
    import primalcr_amd as pcr
    args, torch, dist, rank, N = job.args, job.torch, job.dist, job.rank, job.N
    first, total = shard
    p = pcr.Parameter(k=r, precision=prec, device=job.device, do_predict=0, maxiter=1, **{"lambda": lam})
    t_create = time.perf_counter()
    with pcr.tuned(**({"count_rows": 1} if count_rows else {})):
        s = pcr.Solver(ds, p, rank, N, shard=shard)
    t_create = time.perf_counter() - t_create          # pcr_solver_create(_shard): uploads + the set-up built on the device
    comm_used = None
    if N > 1:
        p2p_name = shm_name + ("_64" if prec == pcr.PCR_F64 else "_32") + ("c" if count_rows else "")
        comm_used = args.comm
        if args.comm == "p2p":
            s.comm_init_p2p(p2p_name)
        else:
            # RCCL -- and if ANY rank cannot join its communicator (an error code from ncclCommInitRank: a layout RCCL refuses, a
            # transport it cannot set up), the WHOLE job takes the direct peer-to-peer exchange instead of ending without a
            # measurement: the ranks agree through the launcher's process group, every rank starts over with a fresh solver.
            err = None
            try:
                s.comm_init(job.bcast(pcr.comm_unique_id() if rank == 0 else None))
            except Exception as e:
                err = f"{type(e).__name__}: {e}"[:200]
            if job.allsum(1.0 if err else 0.0) > 0:
                log(f"[rank {rank}] the RCCL communicator could not be set up on every rank ({err or 'a peer failed'}): the job takes the peer-to-peer exchange")
                s.close()
                with pcr.tuned(**({"count_rows": 1} if count_rows else {})):
                    s = pcr.Solver(ds, p, rank, N, shard=shard)
                s.comm_init_p2p(p2p_name + "_fb")
                comm_used = "p2p (fallback: the RCCL communicator could not be set up)"
        if s.comm_nranks() != N:                 # "RCCL saw N ranks": nothing is timed on a communicator of another size
            raise RuntimeError(f"the {comm_used} communicator reports {s.comm_nranks()} ranks, the job has {N}")
    # the reference's init stream (util.cpp:80): this rank's rows of initial(d1, k), and V = initial(d2, k)
    s.set_factors_local(pcr.initial_rows(total, r, first, s.n_users), pcr.initial(d2, r))

    def barrier():
        s.sync()
        torch.cuda.synchronize()
        if N > 1:
            dist.barrier()

    # Cold start: iterations 1..min(5, W) of the warm-up, straight from pcr_initial, on the same clock as the timed steps (the
    # sorts' nearly-sorted fast path has nothing to start from yet; the reference's default run is -t 10 from cold).  One
    # untimed iteration first, then back to the initial point: the first launch of every kernel loads its code object.
    cold = None
    if N > 1:
        dist.barrier()              # ranks leave their share of the initial() stream seconds apart: meet before the first exchange
    if warmup >= 1 and cold_start:
        s.iterate(1)
        s.set_factors_local(pcr.initial_rows(total, r, first, s.n_users), pcr.initial(d2, r))
    n_cold = min(5, warmup)
    barrier()
    tc = time.perf_counter()
    objs = [rec["obj"] for rec in s.iterate(n_cold)]
    barrier()
    if n_cold:
        cold = {"iterations": n_cold, "ms_per_step": 1e3 * job.allmax(time.perf_counter() - tc) / n_cold}
    objs += [rec["obj"] for rec in s.iterate(warmup - n_cold)]
    prof_period = 0
    if profile:
        # sampled: every n-th launch of each kernel carries an event pair (an event pair costs ~3 us of queue time: every 4th
        # launch adds 6.6 % to the timed region, every 16th 1.9 %, with the same per-kernel averages)
        prof_period = args.profile_period if args.profile_period > 0 else max(1, min(16, 11 * steps // 12))
        s.profile(True, period=prof_period)
        s.profile_reset()
    barrier()
    rows0 = s.counter("ustep_row_gathers")
    t0 = time.perf_counter()
    inner = {"cg_v": 0, "ls_v": 0, "cg_u": 0, "ls_u": 0}
    # exactly K steps = K outer iterations (V step + U step) of the training loop, as pcr_train runs them (pcr_iterate)
    for rec in s.iterate(steps):
        objs.append(rec["obj"])
        for key in inner:
            inner[key] += rec[key]
    barrier()
    secs = job.allmax(time.perf_counter() - t0)
    prof = s.profile_all() if profile else {}
    launches = {name: s.profile_launches(name) for name in prof}
    scope = {name: s.profile_scope(name) for name in prof}
    s.profile(False)
    rows_by_class = s.class_row_gathers() if count_rows else {}
    te_err, te_ndcg = s.evaluate(1, 10)           # the quality after warmup + steps iterations (before the replay below)
    tr_err, tr_ndcg = s.evaluate(0, 10)
    comm_n, shard_now = s.comm_nranks(), (s.first_user, s.n_users, s.nnz_local)
    # what the event pairs cost the timed region: the same K steps once more, straight after, without them (same clock)
    noev = None
    if profile and prof_period:
        barrier()
        t1 = time.perf_counter()
        s.iterate(steps)
        barrier()
        noev = job.allmax(time.perf_counter() - t1)
    # the HBM side: >= 1 s of whole iterations back to back with the memory controllers' activity sampled beside them (HbmSampler)
    hb = None
    if args.hbm and N == 1 and not count_rows and steps > 0:
        smp = HbmSampler(job.device)
        if smp.available():
            # The replay repeats THE SAME iterations as the measurement (warm-up + timed steps from pcr_initial, again and again for
            # >= 1 s) -- not hundreds of further ones: past convergence an fp32 line search no longer finds a decrease and runs
            # its 20 halvings, which is another workload (the first version of this replay ran 665 more ml1m iterations at 2.64 ms
            # each instead of 1.50).  The reset is one upload of the initial factors per block (ml1m: 1 ms in 37).
            U0, V0 = pcr.initial_rows(total, r, first, s.n_users), pcr.initial(d2, r)
            blk = max(1, warmup + steps)
            n_blk = max(1, int(1.0 / max(blk * secs / steps, 1e-6)) + 1)
            n_rep = n_blk * blk
            barrier()
            time.sleep(0.05)                              # (the accumulator ticks in milliseconds: start from a quiet device)
            a0 = smp.mem_activity_acc()
            smp.start()
            t2 = time.perf_counter()
            for _ in range(n_blk):
                s.set_factors_local(U0, V0)
                s.iterate(blk)
            barrier()
            dt = time.perf_counter() - t2
            bp, ns = smp.stop()
            time.sleep(0.01)
            a1 = smp.mem_activity_acc()
            src = "sampled"
            if a0 is not None and a1 is not None and a1 >= a0:
                bp, src = (a1 - a0) / (1e3 * dt), "accumulated"       # percent x ms over the replay's milliseconds
            hb = dict(busy_percent=bp, samples=ns, steps=n_rep, secs=dt, source=src)
    out = dict(secs=secs, secs_noevents=noev, create_s=t_create, objs=objs, inner=inner, prof=prof, launches=launches, scope=scope, prof_period=prof_period, cold=cold, hbm=hb,
               u_rows=s.counter("ustep_row_gathers") - rows0, rows_by_class=rows_by_class, steps=steps,
               te=(te_err, te_ndcg), tr=(tr_err, tr_ndcg), comm_nranks=comm_n, shard=shard_now, comm_used=comm_used)
    s.close()
    return out


def analyse(run, rows_run, wl, prec_name, N, traffic_key, verbose=False, live=None, live_iters=0):
    """Roofline blocks of one timed run.  wl: dict(d1, d2, nnz, r) of the JOB; run["shard"] is this rank's part."""
    secs, inner, prof, steps = run["secs"], run["inner"], run["prof"], run["steps"]
    d1, d2, nnz, r = wl["d1"], wl["d2"], wl["nnz"], wl["r"]
    esz = 4 if prec_name == "f32" else 8
    ld = (r + 3) & ~3
    nu_loc, nnz_loc = run["shard"][1], run["shard"][2]
    # which level of the hierarchy serves the row gathers: the per-user kernels and k_sddmm gather rows of the
    # item table V (d2 x ld x esz); k_spmm (and k_sddmm in its tile-major form, item tables beyond the L2s) gathers rows of U
    # from a user tile cut to fit one XCD's L2 (<= 1.25 MB)
    v_table = d2 * ld * esz
    lvl_v, ceil_v, l2share_v = gather_ceiling(v_table)
    tiled = lambda cls: cls == "spmm" or (cls == "sddmm" and v_table > (32 << 20))         # gathers from an L2-sized user tile

    def binding(cls, gb):
        lv, ce, h = ("l2-gather", GATHER_CEILING_GBS["l2-gather"], 1.0) if tiled(cls) else (lvl_v, ceil_v, l2share_v)
        # gb is None: the rows were not counted (--no-rows) -- an unmeasured figure is null, never 0.0
        return {"level": lv, "l2_share": h, "ceiling_GBs": ce, "achieved_GBs": None if gb is None else round(gb, 1),
                "frac": None if gb is None else round(gb / ce, 4)}
    u_rows = rows_run["u_rows"] if rows_run else None             # None: the U step's row counter was off (--no-rows)
    rows_by_class = (rows_run or {}).get("rows_by_class", {})
    roof, roof_phase, kernels = None, {}, {}
    if prof:
        traffic, traffic_src = {}, None
        tpath = os.path.join(ROOT, "profiles", TRAFFIC_FILE)      # PMC passes of this command (tools/pmc_traffic.py)
        if os.path.exists(tpath) and traffic_key:
            tj = json.load(open(tpath)).get("workloads", {}).get(traffic_key)      # taken on exactly this workload, or absent
            if tj:
                traffic, traffic_src = tj.get("kernels", {}), tj.get("source")
        if live:                                                  # measured on this box by this run (live_traffic)
            traffic, traffic_src = live, "live"
        # Launches are SAMPLED, so a slot's time in the region is its average times ALL its launches.
        est = {name: ((ms / n) * max(run["launches"][name], n) if n else 0.0) for name, (ms, n) in prof.items()}
        total_ms = sum(v for k, v in est.items() if not k.startswith("wall:"))
        for name, (ms, n) in prof.items():
            cls, _, tag = name.partition("/")
            if cls not in SLOT_KERNEL or n == 0:
                continue
            nnz_b, nu_b = run["scope"][name]              # ratings / users one launch of this slot covers
            ab = algorithmic_bytes(name, nnz_b, nu_b, d2, r, esz)
            avg_s = ms / n / 1e3
            tr = None
            # (every k_ustep length class has a kernel symbol of its own -- template parameter CLS -- so a per-symbol PMC
            # average belongs to one slot; a slot that still shares its symbol with another one carries no traffic figure)
            shared = sum(1 for other in prof if other != name and prof[other][1] and other.partition("/")[0] == cls and
                         any(slot_kernel_match(other, kn, prec_name) and slot_kernel_match(name, kn, prec_name) for kn in traffic))
            for kname, t in traffic.items():
                if not shared and slot_kernel_match(name, kname, prec_name):
                    tr = traffic_bytes(t)
            # row gathers of one launch: one ld*esz-byte factor row per rating and half-pass
            g_rows = {"sddmm": nnz_b, "spmm": nnz_b}.get(cls)
            if cls == "ustep" and name in rows_by_class and run["launches"][name]:
                g_rows = rows_by_class[name] / max(rows_run["launches_counted"].get(name, 0), 1)
            kernels[name] = {"avg_us": round(avg_s * 1e6, 2), "timed_launches": int(n), "launches": int(run["launches"][name]),
                             "gpu_time_share": round(est[name] / total_ms, 4), "concurrent_group": cls in ("ustep", "prepare", "eval"),
                             "algorithmic_bytes": int(ab), "achieved_GBs": round(ab / avg_s / 1e9, 2),
                             "frac_hbm_peak": round(ab / avg_s / 1e9 / HBM_PEAK_GBS, 5), "traffic_bytes": tr,
                             "traffic_over_algorithmic": round(tr / ab, 2) if tr else None}
            if g_rows:
                kernels[name]["binding"] = dict(binding(cls, g_rows * r * esz / avg_s / 1e9), gathered_row_bytes=int(g_rows * r * esz))
        if verbose:
            for k, (ms, n) in sorted(prof.items(), key=lambda kv: -est[kv[0]]):
                extra = f"  alg {kernels[k]['achieved_GBs']:8.1f} GB/s  gpu-time share {100 * kernels[k]['gpu_time_share']:5.1f} %" if k in kernels else ""
                log(f"  {k:16s} {est[k]:9.3f} ms  {n:6d} timed  {1e3 * ms / max(n, 1):9.1f} us/launch{extra}")
        # dominant kernel = the slot with the most GPU time (average duration x launches), the way `rocprofv3 --stats` ranks
        # kernels -- concurrent length classes are NOT discounted for running side by side
        dom = max(kernels, key=lambda k: kernels[k]["gpu_time_share"])
        kd = kernels[dom]
        roof = {"bound": "hbm", "kernel": dom, "achieved": kd["achieved_GBs"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": kd["frac_hbm_peak"], "traffic": kd["traffic_bytes"], "traffic_over_algorithmic": kd["traffic_over_algorithmic"],
                "avg_launch_us": kd["avg_us"],
                "launches_timed": kd["timed_launches"], "algorithmic_bytes_per_launch": kd["algorithmic_bytes"],
                "share_of_gpu_time": kd["gpu_time_share"], "binding": kd.get("binding"),
                "traffic_source": (None if kd["traffic_bytes"] is None else
                                   f"live: one rocprofv3 --pmc {PMC_READ} {PMC_WRITE} pass started by this run on this box (a child "
                                   "process replaying the workload), per launch, 32 B x (reads + writes): the L2s' requests to local "
                                   "memory, byte-exact (profiles/r06_dram_calib.md), Infinity-Cache hits INCLUDED -- the HBM side is "
                                   "'hbm'" if traffic_src == "live" else
                                   (traffic_src or f"profiles/{TRAFFIC_FILE}") + ": stored rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this "
                                   "command (separate runs), per launch, 2 x FETCH_SIZE + WRITE_SIZE (MI355X_MICROARCH.md); not "
                                   "measured by this run"),
                "note": "dominant = largest GPU time (average launch duration x launches, HIP events on the launch stream), as "
                        f"rocprofv3 --stats ranks kernels; every {run['prof_period']}th launch of a kernel is event-timed (once-per-step "
                        "kernels every 4th). 'frac' prices the launch's ALGORITHMIC bytes against the HBM peak (the north star's "
                        "yardstick); 'binding' names the level that serves this kernel's row gathers on this shape (by the size of "
                        "the gathered table), its measured ceiling (MI355X_MICROARCH.md, Indexed rows) and the fraction of that; all "
                        "slots in 'kernels', phases in 'roofline_phase' (DESIGN.md 3.5, 4)"}

        # phases: algorithmic bytes of everything a phase launches per step / its wall time per step
        def phase(names, wall_ms_per_step, gather_bytes, cls):
            ab = sum(kernels[k]["algorithmic_bytes"] * run["launches"][k] / steps for k in names)
            gb = None if gather_bytes is None else gather_bytes / (wall_ms_per_step / 1e3) / 1e9
            return {"bound": "hbm", "algorithmic_bytes_per_step": int(ab), "wall_us_per_step": round(1e3 * wall_ms_per_step, 1),
                    "achieved": round(ab / (wall_ms_per_step / 1e3) / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(ab / (wall_ms_per_step / 1e3) / 1e9 / HBM_PEAK_GBS, 5),
                    "share_of_step": round(wall_ms_per_step / (1e3 * secs / steps), 4),
                    "gathered_row_bytes_per_step": None if gb is None else int(gather_bytes),
                    "gather_GBs": None if gb is None else round(gb, 1), "binding": binding(cls, gb)}
        un = [k for k in kernels if k.startswith("ustep/")]
        if un and prof.get("wall:ustep", (0, 0))[1]:
            wm, wn = prof["wall:ustep"]
            # pcr_tune("ustep_newton"): k_unewton runs on the solver's stream BEFORE the classes fork -- it is U-step time (round 5's
            # line booked it under the V step, which is "the rest of the step")
            nm, nn = prof.get("unewton", (0.0, 0))
            newton_ms = (nm / nn) * run["launches"].get("unewton", 0) / steps if nn else 0.0
            u_gather = None if u_rows is None else u_rows / steps / N * r * esz     # this rank's share (the counter is the all-rank total)
            roof_phase["u_step"] = dict(phase(un, wm / wn + newton_ms, u_gather, "ustep"), kernels=un, exact_newton_us_per_step=round(1e3 * newton_ms, 1) if nn else None,
                                        note="all length classes of k_ustep, launched side by side: sum of their algorithmic bytes / fork..join wall "
                                             "time on the solver's stream; gather_GBs = rows of V actually gathered (counted in the kernel: per user "
                                             "1 + 2 per CG iteration + 1 per line-search try, x its ratings) x row bytes / that wall time, against "
                                             "the ceiling of the level that serves the item table of this shape (binding)")
        vn = [k for k in kernels if k.partition("/")[0] in ("sddmm", "spmm", "spmm_fin", "vhv", "vgrad", "cg", "prepare")]
        if vn:
            u_wall = roof_phase.get("u_step", {}).get("wall_us_per_step", 0.0) / 1e3
            v_ms = 1e3 * secs / steps - u_wall               # the two half steps alternate on one stream: the rest of a step is the V step
            v_gather = (inner["ls_v"] / steps + 2 * (inner["cg_v"] / steps) + 1) * esz * r * nnz_loc
            roof_phase["v_step"] = dict(phase(vn, v_ms, v_gather, "sddmm"), kernels=vn,
                                        note="gradient + CG (SDDMM, sweep, SpMM, finish, vector update) + line search, back to back on the "
                                             "solver's stream: step time minus the U step's wall time; gather_GBs = (1 SpMM + n_cg x (SDDMM + "
                                             "SpMM) + n_ls SDDMM) x ratings x row bytes / that time (the SpMM's user tiles are cut to one "
                                             "XCD's L2; the SDDMM gathers the item table, or user tiles when the item table is beyond the L2s)")
    # The HBM side of the roofline (north star: "achieved-HBM-GB/s against gfx950 peak").  No rocprofv3 counter of this part sees
    # past the Infinity Cache (profiles/r06_dram_calib.md); the memory controllers' activity does (HbmSampler): sampled beside a
    # sustained replay of whole iterations straight after the timed region.  fabric = what the L2s asked local memory for per
    # iteration (the live counter pass: every kernel of the training step, Infinity-Cache hits included).
    hbm, hb = None, run.get("hbm")
    fabric_it = None
    if live and live_iters:
        fabric_it = sum(traffic_bytes(t) * t["launches"] for k, t in live.items() if HOT_KERNEL.match(k)) / live_iters
    if hb and hb.get("busy_percent") is not None:
        rate = hb["busy_percent"] * HBM_GBS_PER_BUSY_PERCENT
        per_it = rate * 1e9 * hb["secs"] / hb["steps"]
        hbm = {"busy_percent": round(hb["busy_percent"], 2), "achieved_GBs": round(rate, 1), "peak": HBM_PEAK_GBS, "frac": round(rate / HBM_PEAK_GBS, 5),
               "bytes_per_iteration": int(per_it), "samples": hb["samples"], "replay_steps": hb["steps"], "replay_ms_per_step": round(1e3 * hb["secs"] / hb["steps"], 4),
               "resolution_GBs": HBM_GBS_PER_BUSY_PERCENT if hb.get("source") != "accumulated" else round(HBM_GBS_PER_BUSY_PERCENT / (1e3 * hb["secs"]), 3),
               "source": hb.get("source", "sampled"),
               "fabric_bytes_per_iteration": None if fabric_it is None else int(fabric_it),
               "mall_served_frac": None if not fabric_it else round(max(0.0, 1.0 - per_it / fabric_it), 4),
               "algorithmic_bytes_per_iteration": None,
               "method": "the SMU's average memory-controller activity of this GPU beside "
                         f"{hb['steps']} iterations (the measurement's own warm-up + timed steps from the initial factors, repeated for >= 1 s) -- rocm_smi's accumulating 'Memory Activity' counter (percent x ms) read before and "
                         "after the replay ('accumulated'), else /sys/class/drm/card*/device/mem_busy_percent sampled every 20 ms ('sampled', whole "
                         f"percent) -- x {HBM_GBS_PER_BUSY_PERCENT:g} GB/s per percent (calibrated on streamed "
                         "reads / writes / copies of 1 GiB and on Infinity-Cache-resident re-reads, which read 0 %: profiles/r06_umc_calib.md); "
                         "fabric_bytes = the L2s' requests to local memory per iteration from the live counter pass, Infinity-Cache hits "
                         "included; mall_served_frac = 1 - HBM / fabric"}
        if roof:
            roof["hbm_achieved_GBs"] = hbm["achieved_GBs"]
            roof["hbm_bytes"] = int(rate * 1e9 * roof["avg_launch_us"] * 1e-6)       # HBM bytes delivered chip-wide during one launch's duration
            roof["mall_served_frac"] = hbm["mall_served_frac"]
            roof["hbm_note"] = ("hbm_achieved_GBs: what the memory controllers delivered while whole iterations ran back to back (see 'hbm'); "
                                "hbm_bytes = that rate x this kernel's average launch duration -- an upper bound of its own share, the "
                                "length classes run side by side; mall_served_frac: of the bytes the L2s asked local memory for per "
                                "iteration, the part the Infinity Cache served")
    passes = (1 + (inner["cg_v"] + inner["ls_v"]) / steps) + (1 + (inner["cg_u"] + inner["ls_u"]) / steps / max(d1, 1))
    # SURVEY 8d, the whole-iteration figure: compulsory bytes W of one outer iteration with ideal caching (esz-byte factors,
    # int32 item, uint8 level, esz-byte m, uint32 permutation) at the EXECUTED inner counts, over the measured time per
    # iteration -- all ranks' bytes over the job's time, against N x 8 TB/s.
    n_cg, n_ls = inner["cg_v"] / steps, inner["ls_v"] / steps
    B_csr, F_U, F_V = 5 * nnz + 8 * (d1 + 1), esz * r * d1, esz * r * d2 * N      # V is replicated on every rank
    P_m, P_sort = B_csr + F_U + F_V + esz * nnz, 8 * nnz
    P_hv, P_obj, P_u = B_csr + 8 * nnz + F_U + 2 * F_V, B_csr + 8 * nnz, B_csr + esz * nnz + 2 * F_U + F_V
    W = P_m + P_sort + P_hv + n_cg * P_hv + P_obj + n_ls * (P_m + P_sort + P_obj) + P_u
    it_roof = {"bound": "hbm", "algorithmic_bytes_per_iteration": int(W), "achieved": round(W / (secs / steps) / 1e9, 2),
               "peak": HBM_PEAK_GBS * N, "unit": "GB/s", "frac": round(W / (secs / steps) / 1e9 / (HBM_PEAK_GBS * N), 5),
               "note": "SURVEY 8d: W = P_m + P_sort + P_g + n_cg P_Hv + P_obj + n_ls (P_m + P_sort + P_obj) + P_U at the executed "
                       "n_cg, n_ls; where the factor tables are cache-resident the path is gather/latency-bound and "
                       "this fraction is small by construction (DESIGN.md 3.5) -- see 'gather' for the level that binds"}
    # SURVEY 8d, secondary (diagnostic) figure: row-gather bytes.  One SDDMM or SpMM half-pass moves G = esz * r bytes per
    # rating; per outer iteration the V side makes (1 + n_ls) SDDMMs of the prepares + n_cg of the CG + (1 + n_cg) SpMMs, the
    # U side per rating 1 (gradient) + 2 per CG iteration + 1 per line-search try.
    G = esz * r * nnz
    n_cg_u, n_ls_u = inner["cg_u"] / steps / max(d1, 1), inner["ls_u"] / steps / max(d1, 1)
    v_passes = (n_ls + n_cg) + (1 + n_cg)
    if u_rows is None:                                 # --no-rows: the U side's passes are unknown -- no whole-iteration figure
        u_half_passes = gather_passes = gg = None
    else:
        u_half_passes = u_rows / steps / max(nnz, 1)   # counted in k_ustep: rating-weighted, not user-averaged
        gather_passes = v_passes + u_half_passes
        gg = gather_passes * G / (secs / steps) / 1e9
    rnd = lambda x, n: None if x is None else round(x, n)
    gather = {"bytes_per_half_pass": int(G), "half_passes_per_iteration": rnd(gather_passes, 2), "v_side_half_passes": round(v_passes, 2),
              "u_side_half_passes": rnd(u_half_passes, 2), "u_side_user_average": round(1 + 2 * n_cg_u + n_ls_u, 2),
              "u_side_counted": u_rows is not None,
              "achieved_GBs": rnd(gg, 1), "item_table_bytes": int(v_table), "level": lvl_v, "l2_share": l2share_v,
              "ceiling_GBs": ceil_v * N, "frac": None if gg is None else round(gg / (ceil_v * N), 4),
              "note": "row gathers (one esz*r-byte factor row per rating and half-pass) sustained over the WHOLE iteration, all "
                      "ranks, against the ceiling of the level the item table of this shape lives in (U side: rows counted by the "
                      "kernel -- long users run more CG iterations than the user average, so the rating-weighted pass count is the "
                      "higher one)"}
    # N > 1: what one step spends in its exchange steps (the "allreduce" slot: event pairs around ncclAllReduce / the
    # peer-to-peer exchange on the stream they are queued on) -- so that a scaling curve can be decomposed
    exchange = None
    if N > 1 and prof.get("allreduce", (0, 0))[1]:
        ms, n = prof["allreduce"]
        per_step = run["launches"]["allreduce"] / steps
        exchange = {"allreduce_us_avg": round(1e3 * ms / n, 2), "allreduces_per_step": round(per_step, 2), "timed": int(n),
                    "vector_bytes": int(d2 * ld * esz), "vector_allreduces_per_step": round(1 + n_cg, 2),
                    "scalar_allreduces_per_step": round(per_step - (1 + n_cg) * wl.get("n_rng", 1), 2),
                    "us_per_step": round(1e3 * ms / n * per_step, 1), "share_of_step": round(ms / n * per_step / (1e3 * secs / steps), 4)}
    if hbm:
        hbm["algorithmic_bytes_per_iteration"] = int(W)
    return dict(roofline=roof, roofline_phase=roof_phase, roofline_iteration=it_roof, gather=gather, kernels=kernels,
                passes_per_step=passes, exchange=exchange, hbm=hbm)


def measure(job, shape, r, lam, steps, warmup, users=None, nnz=None, precisions=("f32", "f64"), profile=True, cpu=True,
            cpu_sample=2_000_000, cpu_single=True, verbose=False, cli=None):
    """Generate this rank's shard of `shape`, run the timed legs, return rank 0's record (None on the other ranks)."""
    import primalcr_amd as pcr
    from primalcr_amd import synth
    args, rank, N = job.args, job.rank, job.N
    t0 = time.time()
    R, first, total, scaling, nnz_job = make_shard(shape, job, users, nnz)
    ds = pcr.Dataset.from_ratings(R)
    n_pairs = int(job.allsum(float(ds.count_pairs())))
    tnnz = int(job.allsum(float(len(R.tval))))
    note = {"ml1m": f"ml1m-shaped PrimalCR++ -k {r} -l {lam:g} (configs[1])",
            "netflix": f"Netflix-shaped PrimalCR++ -k {r} -l {lam:g} (configs[3])",
            "yahoo": f"Yahoo!Music-shaped PrimalCR++ -k {r} -l {lam:g} (configs[4]" +
                     ("" if total == synth.SHAPES["yahoo"][0] else f": first {total} of 1.8 M users") + ")"}[shape]
    if rank == 0:
        log(f"[data] {shape}-shaped x{N}: {total} users x {R.d2} items, {nnz_job} ratings, {n_pairs} ordered pairs, "
            f"{tnnz} test ratings; this rank: users [{first}, {first + R.d1}), {R.nnz} ratings ({time.time() - t0:.1f}s)")
    shm = job.bcast(f"/pcr_bench_{os.getpid()}_{int(time.time()) % 100000}_{shape}")
    wl = dict(d1=total, d2=R.d2, nnz=nnz_job, r=r)
    shard = (first, total)
    runs = {}
    for pn in precisions:
        prec = pcr.PCR_F32 if pn == "f32" else pcr.PCR_F64
        runs[pn] = timed_run(job, ds, shard, R.d2, r, lam, prec, steps, warmup, profile, shm)
        # diagnostic replay of the same iterations with the U-step kernels counting the rows of V they gather (an extra atomic
        # per user, so it is kept out of the timed run): the rating-weighted pass count of the U step, per length class
        runs[pn]["rows"] = None
        if not args.no_rows:
            rr = timed_run(job, ds, shard, R.d2, r, lam, prec, steps, warmup, False, shm, count_rows=True, cold_start=False)
            runs[pn]["rows"] = dict(u_rows=rr["u_rows"], rows_by_class=rr["rows_by_class"],
                                    launches_counted={k: (warmup + steps) for k in rr["rows_by_class"]})
    bounds = None
    if N > 1:
        import torch
        t = torch.zeros(N, 3, dtype=torch.float64, device="cuda" if args.rendezvous == "nccl" else "cpu")
        t[rank] = torch.tensor([float(x) for x in runs[precisions[0]]["shard"]], dtype=torch.float64)
        job.dist.all_reduce(t)
        bounds = [[int(v) for v in row] for row in t.tolist()]
    if rank != 0:
        return None
    main_p = precisions[0]
    run = runs[main_p]
    secs, objs, inner = run["secs"], run["objs"], run["inner"]
    whole_yahoo = shape == "yahoo" and users == synth.SHAPES["yahoo"][0] and nnz is None          # configs[4] itself on one GPU
    tkey = (f"{shape}{'-whole' if whole_yahoo else ''}:{main_p}"
            if ((users is None or whole_yahoo) and nnz is None and N == 1 and r == (200 if shape == "yahoo" else 100)) else None)
    live, live_iters = None, 0
    if args.live_traffic and N == 1 and profile and tkey:
        lt_steps, lt_warm = (10, 5) if shape == "ml1m" else (3, 1)
        live, live_iters, took = live_traffic(shape, main_p, r, lt_steps, lt_warm, timeout_s=170 if shape == "ml1m" else (1500 if whole_yahoo else 600),
                                              users=users if whole_yahoo else None)
        log("[traffic] live rocprofv3 --pmc pass: " + (f"{took:.0f} s, {live_iters} iterations replayed" if live else f"not available ({took}): stored passes used"))
    an = analyse(run, run["rows"], wl, main_p, N, tkey, verbose, live, live_iters)
    value = n_pairs * steps / secs
    rec = {
        "value": value, "unit": "pairs/s", "ms_per_step": 1e3 * secs / steps, "s_per_iter": secs / steps, "steps": steps, "warmup": warmup,
        # the timed region carries the HIP-event pairs the roofline's durations come from; the same K steps straight after, without them:
        "ms_per_step_noevents": None if not run.get("secs_noevents") else 1e3 * run["secs_noevents"] / steps,
        "profile_overhead_pct": None if not run.get("secs_noevents") else 100.0 * (secs / run["secs_noevents"] - 1.0),
        "solver_create_s": run.get("create_s"),          # rank 0's pcr_solver_create: off the timed region (the reference's convert(), util.cpp:219-274)
        "dtype": main_p, "scaling": scaling,
        "workload": f"{note}; {total} users x {R.d2} items, {nnz_job} ratings, {n_pairs} ordered pairs; 1 step = 1 outer "
                    f"iteration (V step + U step)",
        "ndcg10_test": run["te"][1], "pairwise_error_test": run["te"][0], "ndcg10_train": run["tr"][1], "pairwise_error_train": run["tr"][0],
        "outer_iterations_run": warmup + steps, "objective": objs[-1],
        "inner_per_step": {k: v / steps for k, v in inner.items()},
        # SURVEY 8d, kernel-level figure: ordered pairs swept per second over the EXECUTED sweep passes of a step
        # (V side: gradient + Hessian-vector products + line-search objectives; U side the same per user, averaged)
        "passes_per_step": an["passes_per_step"], "sweep_pairs_per_s": value * an["passes_per_step"],
        "comm_nranks": run["comm_nranks"], "comm_used": run.get("comm_used"), "shards": bounds, "exchange_profile": an["exchange"],
        "cold_start": run["cold"], "ms_per_step_first5": run["cold"]["ms_per_step"] if run["cold"] and run["cold"]["iterations"] == 5 else None,
        "roofline": an["roofline"], "roofline_phase": an["roofline_phase"], "roofline_iteration": an["roofline_iteration"],
        "gather": an["gather"], "kernels": an["kernels"], "hbm": an["hbm"],
    }
    if "f64" in runs and main_p == "f32":
        # the reference computes in fp64 throughout: the same K steps with fp64 storage, same clock, same barriers
        r64 = runs["f64"]
        live64, live64_iters = None, 0
        if live and shape == "ml1m":               # (the fp32 pass worked on this box: the same pass for the fp64 leg)
            live64, live64_iters, took = live_traffic(shape, "f64", r)
            log(f"[traffic] live rocprofv3 --pmc pass, fp64 leg: " + (f"{took:.0f} s" if live64 else f"not available ({took}): stored passes used"))
        a64 = analyse(r64, r64["rows"], wl, "f64", N, f"{shape}:f64" if tkey else None, live=live64, live_iters=live64_iters)
        rec["f64"] = {"dtype": "f64", "ms_per_step": 1e3 * r64["secs"] / steps,
                      "ms_per_step_noevents": None if not r64.get("secs_noevents") else 1e3 * r64["secs_noevents"] / steps, "value": n_pairs * steps / r64["secs"], "cold_start": r64["cold"],
                      "unit": "pairs/s", "ndcg10_test": r64["te"][1], "pairwise_error_test": r64["te"][0],
                      "objective": r64["objs"][-1], "inner_per_step": {k: v / steps for k, v in r64["inner"].items()},
                      "roofline": a64["roofline"], "roofline_phase": a64["roofline_phase"], "roofline_iteration": a64["roofline_iteration"],
                      "gather": a64["gather"], "kernels": a64["kernels"], "hbm": a64["hbm"],
                      "note": "second timed run of the same workload with U, V, m and the CG vectors stored in fp64 (the reference's "
                              "arithmetic type); 'value' above is the fp32-storage / fp64-accumulation run the north star allows "
                              "('within fp32 tolerance')"}
        rec["f64_minus_f32"] = {"ndcg10_test": r64["te"][1] - run["te"][1], "pairwise_error_test": r64["te"][0] - run["te"][0],
                                "objective_rel": r64["objs"][-1] / objs[-1] - 1}
    rec["cpu_baseline"] = None
    if N == 1 and (cpu or cli is not None):
        import shutil
        td = tempfile.mkdtemp(prefix="pcr_data_", dir="/tmp")
        try:
            data_dir = None
            if cli is not None or R.nnz <= cpu_sample:                   # the workload as the reference's text directory (meta + rating files)
                t0 = time.time()
                data_dir = synth.write_dir(R, os.path.join(td, "data"))
                log(f"[data] text directory written in {time.time() - t0:.1f}s")
            if cpu:
                rec["cpu_baseline"] = cpu_baseline(R, n_pairs, r, lam, cpu_sample, cpu_single, data_dir)
                rec["speedup_vs_cpu_baseline"] = value / rec["cpu_baseline"]["value"]
                if "f64" in rec:
                    rec["f64"]["speedup_vs_cpu_baseline"] = rec["f64"]["value"] / rec["cpu_baseline"]["value"]
            if cli is not None:
                try:
                    rec["cli"] = cli_leg(data_dir, r, lam, **cli)
                except Exception as e:               # (a time-out, a missing binary: the leg reports it, the measurement stands)
                    rec["cli"] = {"error": f"{type(e).__name__}: {e}"[:300]}
                c = rec["cli"]
                log("[cli] omp-pmf-train end to end: " + (c.get("error") or
                    f"{c['wall_s']:.2f}s wall (load {c['load_s']:.2f} init {c['init_s']:.2f} create {c['create_s']:.2f} iterations {c['iter_s']:.3f} "
                    f"evaluation {c['eval_s']:.2f} write {c['write_s']:.2f}); reference {c.get('reference_wall_s')}"))
        finally:
            shutil.rmtree(td, ignore_errors=True)
    return rec


LINE_CAP = 5120        # bytes: the driver keeps ~8.6 KB of stdout; round 3's 45 KB line came back unparsed


def _r(x, n=4):
    """Round to n significant-ish digits for the compact line (None stays None)."""
    if x is None or isinstance(x, (bool, str)):
        return x
    if isinstance(x, int):
        return x
    return float(f"{x:.{n}g}")


def _roof(rf):
    """The roofline object the contract asks for, without its prose."""
    if not rf:
        return None
    b = rf.get("binding") or {}
    src = rf.get("traffic_source")
    return {"bound": rf["bound"], "kernel": rf.get("kernel"), "achieved": rf["achieved"], "peak": rf["peak"], "unit": rf["unit"],
            "frac": rf["frac"], "traffic": rf.get("traffic"), "traffic_over_algorithmic": rf.get("traffic_over_algorithmic"),
            "traffic_source": None if not src else ("live-pmc" if src.startswith("live") else "stored-pmc:" + src.split(" ")[0]),
            "avg_launch_us": rf.get("avg_launch_us"), "launches_timed": rf.get("launches_timed"),
            "algorithmic_bytes_per_launch": rf.get("algorithmic_bytes_per_launch"), "share_of_gpu_time": rf.get("share_of_gpu_time"),
            "binding": {"level": b.get("level"), "ceiling_GBs": b.get("ceiling_GBs"), "achieved_GBs": b.get("achieved_GBs"),
                        "frac": b.get("frac")} if b else None,
            "hbm_achieved_GBs": rf.get("hbm_achieved_GBs"), "hbm_bytes": rf.get("hbm_bytes"), "mall_served_frac": rf.get("mall_served_frac")}


def _hbm(h):
    """The HBM side of an iteration, in six numbers."""
    if not h:
        return None
    return {"busy_percent": h.get("busy_percent"), "achieved_GBs": h.get("achieved_GBs"), "frac": h.get("frac"),
            "bytes_per_iteration": h.get("bytes_per_iteration"), "fabric_bytes_per_iteration": h.get("fabric_bytes_per_iteration"),
            "mall_served_frac": h.get("mall_served_frac")}


def _cpu(cb):
    if not cb:
        return None
    o = {"value": _r(cb["value"], 6), "unit": cb["unit"], "cores": cb["cores"], "kind": cb["kind"], "sample": cb["sample"][:200],
         "s_per_iter": _r(cb.get("s_per_iter"), 5)}
    st = cb.get("single_thread")
    if st:
        o["single_thread"] = {"value": _r(st["value"], 6), "cores": 1, "s_per_iter": _r(st.get("s_per_iter"), 5)}
    return o


def _sample_short(sample):
    """What a bounded cpu_baseline was timed on, in a dozen words."""
    m = re.search(r"its first (\d+) users \((\d+) ratings", sample or "")
    return f"first {m.group(1)} users ({m.group(2)} ratings) of the shape" if m else ((sample or "")[:80] or None)


def _phases(rp):
    o = {}
    for k, v in (rp or {}).items():
        o[k] = {"wall_us": v.get("wall_us_per_step"), "share_of_step": v.get("share_of_step"), "frac": v.get("frac"),
                "gather_GBs": v.get("gather_GBs"), "gather_frac": (v.get("binding") or {}).get("frac")}
    return o or None


def _top_kernels(kernels, n=3):
    top = sorted((kernels or {}).items(), key=lambda kv: -kv[1].get("gpu_time_share", 0))[:n]
    return [{"slot": k, "avg_us": v["avg_us"], "share": v["gpu_time_share"], "frac": v["frac_hbm_peak"],
             "traffic_x": v.get("traffic_over_algorithmic")} for k, v in top] or None


def compact_line(full, full_record_path=None):
    """The ONE stdout line of a bench run, built from the full record: the contract's header, `roofline`, `cpu_baseline`, the
    quality figures, the fp64 and Netflix-shaped legs as a few numbers each -- and nothing that grows with the number of
    kernels.  Always below LINE_CAP bytes: optional blocks are dropped, in a fixed order, if a future field pushes it over."""
    g = full.get
    cfg = dict(g("config") or {})
    line = {k: g(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                              "vs_baseline", "dtype", "data")}
    line["config"] = cfg
    line["roofline"] = _roof(g("roofline"))
    line["cpu_baseline"] = _cpu(g("cpu_baseline"))
    line["speedup_vs_cpu_baseline"] = _r(g("speedup_vs_cpu_baseline"))
    for k in ("ndcg10_test", "pairwise_error_test"):
        line[k] = _r(g(k), 6)
    line["objective"] = _r(g("objective"), 9)
    line["ms_per_step_first5"] = _r(g("ms_per_step_first5"), 5)
    line["cold_start"] = {k: _r(v, 5) for k, v in g("cold_start").items()} if g("cold_start") else None
    line["inner_per_step"] = g("inner_per_step")
    line["sweep_pairs_per_s"] = _r(g("sweep_pairs_per_s"), 5)
    line["comm_nranks"] = g("comm_nranks")
    if g("exchange_profile"):
        line["exchange"] = g("exchange_profile")
    if g("shards"):
        line["shards"] = g("shards") if len(g("shards")) <= 8 else None
    it, ga = g("roofline_iteration") or {}, g("gather") or {}
    line["roofline_iteration"] = {"algorithmic_bytes": it.get("algorithmic_bytes_per_iteration"), "achieved": it.get("achieved"),
                                  "frac": it.get("frac")} if it else None
    line["gather"] = {"level": ga.get("level"), "half_passes": ga.get("half_passes_per_iteration"), "achieved_GBs": ga.get("achieved_GBs"),
                      "ceiling_GBs": ga.get("ceiling_GBs"), "frac": ga.get("frac"), "u_side_counted": ga.get("u_side_counted", True)} if ga else None
    line["hbm"] = _hbm(g("hbm"))
    line["roofline_phase"] = _phases(g("roofline_phase"))
    line["top_kernels"] = _top_kernels(g("kernels"))
    line["profile_overhead_pct"] = _r(g("profile_overhead_pct"), 3)
    line["solver_create_s"] = _r(g("solver_create_s"), 3)
    f64 = g("f64")
    if f64:
        # the reference's arithmetic type: the like-for-like leg carries the same blocks as the headline (two numbers per phase)
        rf = _roof(f64.get("roofline")) or {}
        for drop in ("bound", "peak", "unit", "launches_timed", "share_of_gpu_time"):
            rf.pop(drop, None)
        if rf.get("binding"):
            rf["binding"] = {"level": rf["binding"].get("level"), "frac": rf["binding"].get("frac")}
        it64 = f64.get("roofline_iteration") or {}
        line["f64"] = {"ms_per_step": _r(f64["ms_per_step"], 6), "value": _r(f64["value"], 7), "ndcg10_test": _r(f64.get("ndcg10_test"), 6),
                       "ms_per_step_first5": _r((f64.get("cold_start") or {}).get("ms_per_step"), 5),
                       "speedup_vs_cpu_baseline": _r(f64.get("speedup_vs_cpu_baseline")),
                       "roofline": rf or None,
                       "roofline_phase": {k: {"wall_us": v.get("wall_us_per_step"), "frac": v.get("frac")}
                                          for k, v in (f64.get("roofline_phase") or {}).items()} or None,
                       "roofline_iteration_frac": it64.get("frac"),
                       "gather_frac": (f64.get("gather") or {}).get("frac"), "hbm": _hbm(f64.get("hbm"))}
    nf = g("netflix")
    if nf and nf.get("error"):
        line["netflix"] = {"error": str(nf["error"])[:300]}          # the second leg failed: the headline above stands
    elif nf:
        rf, nf64, ncb = nf.get("roofline") or {}, nf.get("f64") or {}, nf.get("cpu_baseline") or {}
        line["netflix"] = {"workload": "configs[3] Netflix-shaped 480189 x 17770, 100 M ratings, k=100, 1 GPU",
                           "ms_per_step": _r(nf["ms_per_step"], 6), "value": _r(nf["value"], 7), "steps": nf.get("steps"), "warmup": nf.get("warmup"),
                           "solver_create_s": _r(nf.get("solver_create_s"), 3),
                           "ndcg10_test": _r(nf.get("ndcg10_test"), 6), "pairwise_error_test": _r(nf.get("pairwise_error_test"), 6),
                           "roofline": {"kernel": rf.get("kernel"), "frac": rf.get("frac"), "avg_launch_us": rf.get("avg_launch_us"),
                                        "traffic_over_algorithmic": rf.get("traffic_over_algorithmic")} if rf else None,
                           "roofline_iteration_frac": (nf.get("roofline_iteration") or {}).get("frac"),
                           "gather": {"level": (nf.get("gather") or {}).get("level"), "frac": (nf.get("gather") or {}).get("frac")},
                           "hbm": _hbm(nf.get("hbm")),
                           "f64_ms_per_step": _r(nf64.get("ms_per_step"), 6),
                           # (timed on a user prefix of the shape: compare pairs/s, not s_per_iter with ms_per_step)
                           "cpu_baseline": {"value": _r(ncb.get("value"), 6), "cores": ncb.get("cores"), "kind": ncb.get("kind"),
                                            "s_per_iter": _r(ncb.get("s_per_iter"), 5),
                                            "sample": _sample_short(ncb.get("sample"))} if ncb else None,
                           "speedup_vs_cpu_baseline": _r(nf.get("speedup_vs_cpu_baseline"))}
        if (g("n_gpus") or 1) > 1:
            # the strong-scaling leg of an N > 1 line: the same user-sharded job over all ranks, its shards and its exchange steps
            nb = line["netflix"]
            nb["workload"] = nf.get("workload_short") or f"configs[3] Netflix-shaped, k=100, user-sharded x{g('n_gpus')}"
            for drop in ("cpu_baseline", "speedup_vs_cpu_baseline", "f64_ms_per_step", "solver_create_s"):
                nb.pop(drop, None)
            nb.update({"scaling": nf.get("scaling"), "comm_nranks": nf.get("comm_nranks"), "objective": _r(nf.get("objective"), 9),
                       "shards": nf.get("shards") if nf.get("shards") and len(nf["shards"]) <= 8 else None, "exchange": nf.get("exchange_profile")})
    cli = g("cli")
    if cli:
        ref = cli.get("reference") or {}
        line["cli"] = {k: _r(cli.get(k), 4) for k in ("wall_s", "load_s", "init_s", "create_s", "train_s", "iter_s", "eval_s", "write_s",
                                                      "reference_wall_s", "speedup_wall", "largest_phase")}
        line["cli"].update({"ndcg10_test": _r(cli.get("ndcg10_test"), 6), "reference_ndcg10_test": _r(ref.get("ndcg10_test"), 6),
                            "error": cli.get("error")})
        if cli.get("create_split"):                      # the three largest parts of create_s (all of them in the full record)
            top = sorted(cli["create_split"].items(), key=lambda kv: -kv[1])[:4]
            line["cli"]["create_split"] = {k: _r(v, 3) for k, v in top if k != "solver_create_s"}
    line["full_record"] = full_record_path
    # the cap: drop optional blocks (least important first) rather than ever print a line the driver cannot keep
    size = lambda: len(json.dumps(line, separators=(",", ":")))
    if size() >= LINE_CAP and len(cfg.get("workload") or "") > 240:
        cfg["workload"] = cfg["workload"][:240]
    for drop in ("hbm_achieved_GBs", "hbm_bytes", "mall_served_frac"):          # (the f64 leg's roofline repeats them in its own 'hbm')
        if f64 and line.get("f64", {}).get("roofline"):
            line["f64"]["roofline"].pop(drop, None)
    for victim in ("top_kernels", "roofline_phase", "shards", "inner_per_step", "gather", "roofline_iteration", "cold_start", "exchange",
                   "cli", "netflix", "f64"):
        if size() < LINE_CAP:
            break
        line.pop(victim, None)
    return line


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def error_line(msg, **extra):
    """The stdout line of a run that failed: a driver that reads the last line sees WHY there is no measurement."""
    print(json.dumps(dict({"error": msg, "metric": "pairwise-comparisons/sec", "value": None}, **extra), separators=(",", ":")), flush=True)


def spawn_ranks(N):
    """--gpus N > 1 without a launcher: start the N ranks as DIRECT child processes of this one -- which has not imported torch
    and never touches a GPU -- with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT in their environment, and relay
    what rank 0 prints (the children inherit stdout / stderr).  Never an exec.

    No torch.distributed.run in between: its agent process imports torch and opens the device too, so an N-rank job was N + 1
    processes on the card -- which is what ended round 4's 6-rank rehearsal on the one-GPU box without a word (the pool's process
    guard allows 6 and kills the whole command; NOTES.md, round 5).  Every way a rank can stop is reported: exit code or signal on
    stderr, the other ranks are taken down, and the last stdout line is {"error": ...}."""
    import signal
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL / the peer-to-peer exchange across processes need it here
    env.setdefault("OMP_NUM_THREADS", str(max(1, host_cores() // N)))
    env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()), WORLD_SIZE=str(N), LOCAL_WORLD_SIZE=str(N))
    log(f"[bench] --gpus {N}: starting {N} ranks (direct children, rendezvous 127.0.0.1:{env['MASTER_PORT']})")
    kids = []
    for q in range(N):
        kids.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:],
                                     env=dict(env, RANK=str(q), LOCAL_RANK=str(q), GROUP_RANK="0")))
    stop = {"sig": None}

    def on_signal(sig, _frame):                  # the parent's own SIGTERM / SIGINT goes to the ranks: nobody is left on the GPU
        stop["sig"] = sig
        for k in kids:
            if k.poll() is None:
                k.send_signal(signal.SIGTERM)
    old = {sg: signal.signal(sg, on_signal) for sg in (signal.SIGTERM, signal.SIGINT)}
    failed, t_last, t_kill = None, time.time(), None
    try:
        while any(k.poll() is None for k in kids):
            for q, k in enumerate(kids):
                rc = k.poll()
                if rc is not None and rc != 0 and failed is None:
                    try:
                        signame = signal.Signals(-rc).name if rc < 0 else ""
                    except ValueError:
                        signame = "?"
                    why = f"killed by signal {-rc} ({signame})" if rc < 0 else f"exited with code {rc}"
                    failed = (q, rc, why)
                    log(f"[bench] rank {q} {why} (pid {k.pid}): stopping the other ranks")
                    for o in kids:
                        if o.poll() is None:
                            o.send_signal(signal.SIGTERM)
                    t_kill = time.time() + 15.0
            if failed and t_kill and time.time() > t_kill:
                for o in kids:
                    if o.poll() is None:
                        log(f"[bench] rank pid {o.pid} ignored SIGTERM for 15 s: SIGKILL")
                        o.kill()
            if time.time() - t_last > 60.0:      # a heartbeat: a long multi-rank run is not mistaken for a hung one
                t_last = time.time()
                log(f"[bench] waiting for ranks {[q for q, k in enumerate(kids) if k.poll() is None]}")
            time.sleep(0.05)
    finally:
        for sg, h in old.items():
            signal.signal(sg, h)
    if failed is None:
        for q, k in enumerate(kids):             # (a rank that failed after the loop's last look)
            if k.returncode != 0:
                failed = (q, k.returncode, f"exited with code {k.returncode}" if k.returncode > 0 else f"killed by signal {-k.returncode}")
                log(f"[bench] rank {q} {failed[2]} (pid {k.pid})")
                break
    if stop["sig"] is not None and failed is None:
        failed = (-1, 128 + stop["sig"], f"the launcher received signal {stop['sig']}")
    if failed:
        error_line(f"rank {failed[0]} {failed[2]}; see stderr", n_gpus=N, rank=failed[0], returncode=failed[1])
        return 1
    return 0


class SecondLeg:
    """The Netflix-shaped strong-scaling leg of an N > 1 run (configs[3]) must never cost the headline: the ml1m record is complete
    when this leg starts, rank 0 HOLDS it (stdout carries ONE line per run) and prints it whatever happens next --
      * an exception on any rank: the rank leaves a note under `flag` (the ranks of a job share one node) and stops;
      * a rank that hangs (a collective whose peer is gone has no time-out of its own): every rank's watchdog thread sees the note
        or the deadline and ends its process -- rank 0 prints the held line with netflix = {"error": ...} first;
      * SIGTERM (a launcher taking the job down because a peer died): rank 0 prints the held line first.
    Processes end through os._exit(0) on those paths: nothing may wait in a destructor for a peer that will not come."""

    def __init__(self, rank, flag, deadline_s, emit):
        import threading
        self.rank, self.flag, self.deadline, self.emit = rank, flag, time.time() + deadline_s, emit
        self.done = threading.Event()
        self.lock = threading.Lock()
        self.left = False
        self.thread = threading.Thread(target=self.watch, daemon=True)

    def leave(self, why):
        with self.lock:                         # (the watchdog, the signal handler and the main thread: the line is printed once)
            if self.left:
                return
            self.left = True
        log(f"[rank {self.rank}] Netflix-shaped leg: {why}")
        if self.rank == 0:
            self.emit({"error": why})
        sys.stdout.flush(); sys.stderr.flush()
        os._exit(0)

    def note(self, why):
        try:
            with open(self.flag, "a") as f:
                f.write(f"rank {self.rank}: {why}\n")
        except OSError:
            pass

    def watch(self):
        import select, signal
        while not self.done.is_set():
            # (a SIGTERM while the main thread sits in a library call: Python runs its handler only once that call returns, but the
            # C-level handler writes the signal's number to the wake-up pipe at once -- this thread reads it)
            ready, _, _ = select.select([self.pipe_r], [], [], 0.5)
            if ready:
                try:
                    got = os.read(self.pipe_r, 64)
                except OSError:
                    got = b""
                if bytes([signal.SIGTERM]) in got:
                    self.note("the job received SIGTERM during the leg")
                    self.leave("the job received SIGTERM during the leg")
            if os.path.exists(self.flag):
                try:
                    why = open(self.flag).read().strip().split("\n")[0]
                except OSError:
                    why = "a rank failed"
                self.leave(why)
            if time.time() > self.deadline:
                self.note("the leg did not finish within its time budget")
                self.leave(f"not finished within its time budget on rank {self.rank}")

    def run(self, fn):
        import signal
        self.pipe_r, pipe_w = os.pipe()
        os.set_blocking(self.pipe_r, False); os.set_blocking(pipe_w, False)
        old_fd = signal.set_wakeup_fd(pipe_w, warn_on_full_buffer=False)
        old = signal.signal(signal.SIGTERM, lambda sig, frame: self.leave("the job received SIGTERM during the leg"))
        self.thread.start()
        try:
            out = fn()
        except BaseException as e:              # (SystemExit / KeyboardInterrupt included: the held line goes out first)
            import traceback
            traceback.print_exc()
            why = f"{type(e).__name__}: {e}"[:300]
            try:                                # (a peer that failed FIRST has left its note: this rank's own error -- a reset connection,
                first = open(self.flag).read().strip().split("\n")[0]      # a poisoned exchange -- is only the echo of it)
            except OSError:
                first = ""
            if first:
                why = first
            else:
                self.note(why)
            self.leave(why)
        self.done.set()
        self.thread.join()
        signal.signal(signal.SIGTERM, old)
        signal.set_wakeup_fd(old_fd)
        os.close(self.pipe_r); os.close(pipe_w)
        if self.rank == 0 and os.path.exists(self.flag):
            os.unlink(self.flag)
        return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--shape", choices=["ml1m", "netflix", "yahoo"], default="ml1m",
                    help="ml1m: configs[1], 6040 users per GPU (weak scaling); netflix: configs[3], 480189 x 17770, 100 M ratings in "
                         "total, user-sharded over the GPUs (strong scaling); yahoo: configs[4], 225000 users per GPU of the 1.8 M "
                         "(all of them at 8 GPUs)")
    ap.add_argument("--users", type=int, default=None, help="override the user count of the shape (per GPU for ml1m, total for netflix, "
                                                            "length of the user prefix for yahoo)")
    ap.add_argument("--nnz", type=int, default=None, help="override the rating count likewise")
    ap.add_argument("--rank-k", type=int, default=None, help="factor rank (BASELINE: 100; yahoo: 200)")
    ap.add_argument("--lam", type=float, default=5000.0)
    ap.add_argument("--precision", choices=["f32", "f64"], default="f32")
    ap.add_argument("--comm", choices=["rccl", "p2p"], default="rccl", help="N > 1: ncclAllReduce, or the direct peer-to-peer exchange")
    ap.add_argument("--devices", default=None, help="HIP device of every local rank, e.g. 0,0 (rehearsal of the N > 1 path on a box with "
                                                    "fewer GPUs than ranks; needs --comm p2p and --rendezvous gloo: RCCL refuses two ranks on one device)")
    ap.add_argument("--rendezvous", choices=["nccl", "gloo"], default="nccl", help="torch.distributed backend of the barriers and the id broadcast")
    ap.add_argument("--tune", action="append", default=[], help="key=value launch knob (pcr_tune), repeatable -- for A/B runs")
    ap.add_argument("--lib", default=None, help="A/B: load this libprimalcr.so instead of primalcr_amd/lib/libprimalcr.so")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-f64", action="store_true", help="skip the second timed run in the reference's arithmetic type")
    ap.add_argument("--no-netflix", action="store_true", help="skip the Netflix-shaped sub-record (the default N = 1 run; every N > 1 run of the ml1m shape)")
    ap.add_argument("--netflix-users", type=int, default=None, help="N > 1: user count of the Netflix-shaped leg (default: the shape's 480 189)")
    ap.add_argument("--netflix-nnz", type=int, default=None, help="N > 1: rating count of the Netflix-shaped leg (default: 100 M)")
    ap.add_argument("--netflix-budget-s", type=float, default=420.0, help="N > 1: wall-clock bound of the Netflix-shaped leg; past it the line goes out "
                                                                          "with netflix = {\"error\": ...} and the headline intact")
    ap.add_argument("--no-cli", action="store_true", help="skip the end-to-end leg of the default N = 1 run: the drop-in omp-pmf-train (-t 10, "
                                                          "defaults) and the reference binary on the workload's text directory, wall-clocked")
    ap.add_argument("--no-profile", action="store_true", help="do not record per-kernel HIP events in the timed region")
    ap.add_argument("--all-legs", action="store_true", help="N > 1: also the fp64 leg and the diagnostic replay that counts the U step's row gathers "
                                                            "(default for N > 1: the one timed run)")
    ap.add_argument("--no-rows", action="store_true", help="skip the diagnostic replay that counts the rows the U step gathers (profiler passes)")
    ap.add_argument("--profile-period", type=int, default=0,
                    help="event-time every n-th launch of each kernel (the first one included); 0 = as sparse as leaves a dozen "
                         "samples of the most frequent kernel (11 launches per step): min(16, 11 * steps / 12)")
    ap.add_argument("--full-record", default=None, help="where the full record (per-kernel tables, phases, notes) is written; default "
                                                        "bench_full.json next to bench.py.  stdout carries the compact line only")
    ap.add_argument("--no-live-traffic", dest="live_traffic", action="store_false",
                    help="do not start the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) that measure roofline.traffic on this box "
                         "(default N = 1 ml1m run only; the stored passes of profiles/ are used instead)")
    ap.add_argument("--no-hbm", dest="hbm", action="store_false",
                    help="skip the sustained replay (>= 1 s of iterations) beside which the memory controllers' activity is sampled: the HBM side of the roofline")
    ap.add_argument("--full-line", action="store_true", help="developer tools only: print the full record as the stdout line (tens of KB)")
    ap.add_argument("--verbose", action="store_true")
    ap.add_argument("--fault-netflix", type=int, default=None, help="test hook: that rank raises inside the Netflix-shaped leg of an N > 1 run (the line must "
                                                                     "still carry the ml1m headline, with netflix = {\"error\": ...})")
    ap.add_argument("--fault", default=None, help="test hook, 'rank:signal' or 'rank:exit:code': that rank ends itself that way before it touches "
                                                  "a GPU (the launcher must name it, stop the others and print an error line)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world == 1 and "RANK" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))                 # (before torch is imported: this process never initialises a GPU)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    N = world
    if args.fault and int(args.fault.split(":")[0]) == rank:
        f = args.fault.split(":")
        if f[1] == "exit":
            os._exit(int(f[2]))
        os.kill(os.getpid(), int(f[1]))
        time.sleep(30)
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no GPU visible (there is no CPU fallback for the training path)")
    device = int(args.devices.split(",")[local_rank]) if args.devices else local_rank
    torch.cuda.set_device(device)
    dist = None
    if N > 1:
        import torch.distributed as dist
        if args.rendezvous == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", device))
        else:
            dist.init_process_group("gloo")

    if local_rank == 0 and not args.lib and not os.path.exists(os.path.join(ROOT, "primalcr_amd", "lib", "libprimalcr.so")):
        import __graft_entry__                   # clean checkout: compile the product first (no fallback exists)
        __graft_entry__.build()
    if N > 1:
        dist.barrier()                           # (every rank, whether or not it saw the library missing)
    import primalcr_amd as pcr
    if args.lib:
        pcr.use_library(args.lib)
    for kv in args.tune:
        pcr.tune(*kv.split("=", 1))
    if args.devices and len(set(args.devices.split(","))) < len(args.devices.split(",")) and not any(kv.startswith("lanes=") for kv in args.tune):
        # ranks that share a device (a rehearsal on fewer GPUs) share its hardware queues: one stream per rank, no lane probing
        pcr.tune("lanes", "1")
        if rank == 0:
            log("[bench] several ranks share a device: one stream per rank (--tune lanes=1)")
    job = Job(args, torch, dist, rank, N, device)
    r, lam = args.rank_k or (200 if args.shape == "yahoo" else 100), args.lam
    # N > 1: ONE timed run and one communicator per shape (the scaling record needs `value`; the fp64 leg and the row-counting replay are three
    # more solvers and communicators per rank on a path that has never met a peer over xGMI -- ask for them with --all-legs)
    if N > 1 and not args.all_legs:
        args.no_f64 = True
        args.no_rows = True
    precisions = (args.precision,) if (args.no_f64 or args.precision == "f64") else ("f32", "f64")
    default_run = (N == 1 and args.shape == "ml1m" and args.users is None and args.nnz is None and args.rank_k is None and args.precision == "f32")
    rec = measure(job, args.shape, r, lam, args.steps, args.warmup, args.users, args.nnz, precisions, not args.no_profile,
                  not args.no_cpu, verbose=args.verbose, cli={} if (default_run and not args.no_cli) else None)
    # configs[3] in the driver's line: the default N = 1 run also times a few steps of the Netflix-shaped set (north star "Target")
    nf = None
    if (N == 1 and args.shape == "ml1m" and not args.no_netflix and args.users is None and args.nnz is None and args.rank_k is None
            and args.precision == "f32"):
        nf = measure(job, "netflix", 100, lam, 3, 1, None, None, precisions, not args.no_profile, not args.no_cpu,
                     cpu_sample=1_000_000, cpu_single=False)
    out = None
    if rank == 0:
        out = {"metric": "pairwise-comparisons/sec", "value": rec["value"], "unit": "pairs/s", "n_gpus": N, "steps": args.steps,
               "warmup": args.warmup, "ms_per_step": rec["ms_per_step"], "higher_is_better": True, "scaling": rec["scaling"],
               "vs_baseline": None, "dtype": rec["dtype"], "data": "synthetic",
               "config": {"workload": rec["workload"], "solver": "PrimalCR++", "rank": r, "lambda": lam,
                          "parallelism": f"user-sharded x{N}", "accumulation": "f64", "storage": rec["dtype"],
                          "exchange": None if N == 1 else (rec.get("comm_used") or args.comm)}}
        for k, v in rec.items():
            if k not in out and k not in ("workload", "scaling", "dtype", "steps", "warmup"):
                out[k] = v

    def emit(netflix):
        """Rank 0: the full record to its file, the ONE compact line to stdout."""
        if netflix:
            out["netflix"] = netflix
        # The driver reads the LAST stdout line and keeps only a few KB of it: the line is the compact summary (< 5 KB, every
        # field the contract names + roofline + cpu_baseline), the full record (per-kernel tables, phases, notes) goes to a file.
        full_path = args.full_record or os.path.join(ROOT, "bench_full.json")
        try:
            with open(full_path, "w") as f:
                json.dump(out, f, indent=1)
            shown = os.path.relpath(full_path, ROOT) if full_path.startswith(ROOT + os.sep) else full_path
        except OSError as e:
            log(f"[bench] could not write the full record to {full_path}: {e}")
            shown = None
        print(json.dumps(out) if args.full_line else json.dumps(compact_line(out, shown), separators=(",", ":")), flush=True)

    # configs[3] in the N > 1 line too (north star "Target": both shapes at 1, 2, 4 and 8 GPUs): once the weak-scaled ml1m record is
    # complete, the SAME ranks build the Netflix-shaped set cut into N nnz-balanced user ranges, with a communicator of its own, and
    # time 3 steps -- strong scaling.  Guarded (SecondLeg): whatever happens in this leg, the line goes out with the headline intact.
    if N > 1 and args.shape == "ml1m" and not args.no_netflix:
        flag = job.bcast(os.path.join(tempfile.gettempdir(), f"pcr_bench_nf_{os.getpid()}_{int(time.time())}.failed"))
        leg = SecondLeg(rank, flag, args.netflix_budget_s, emit)

        def second():
            if args.fault_netflix is not None and args.fault_netflix == rank:
                raise RuntimeError("fault hook: this rank fails in the Netflix-shaped leg")
            return measure(job, "netflix", 100, lam, 3, 1, args.netflix_users, args.netflix_nnz, ("f32",), not args.no_profile, False)
        nf = leg.run(second)
        if rank == 0 and nf is not None:
            nf["workload_short"] = (f"configs[3] Netflix-shaped {nf['workload'].split('; ')[1]}, k=100, user-sharded x{N}")[:160]
    if rank == 0:
        emit(nf)
    if N > 1:
        dist.barrier(); dist.destroy_process_group()


def guarded_main():
    """Every way a rank can stop says why: `[rank q] ...` with the traceback on stderr, a non-zero exit code, and -- on rank 0,
    whose stdout is the one a driver reads -- {"error": ...} as the last stdout line."""
    rank = os.environ.get("RANK", "0")

    def let_rank0_speak():
        # A launcher (torch.distributed.run) answers the first non-zero exit code with SIGTERM to every other rank: a rank other
        # than 0 that fails for a reason ALL ranks share (no GPU visible, a missing library) waits a moment, so that rank 0's
        # {"error": ...} line -- the one a driver reads -- is out before the launcher takes rank 0 down.
        if rank != "0":
            sys.stdout.flush(); sys.stderr.flush()
            time.sleep(1.5)
    try:
        main()
    except SystemExit as e:
        if e.code not in (None, 0) and not isinstance(e.code, int):
            log(f"[rank {rank}] {e.code}")
            if rank == "0":
                error_line(str(e.code), rank=0)
            let_rank0_speak()
            sys.exit(1)
        raise
    except BaseException as e:                   # (KeyboardInterrupt / SIGTERM-as-exception included)
        import traceback
        log(f"[rank {rank}] failed: {type(e).__name__}: {e}")
        traceback.print_exc()
        if rank == "0":
            error_line(f"{type(e).__name__}: {e}"[:500], rank=0)
        sys.stdout.flush(); sys.stderr.flush()
        let_rank0_speak()
        os._exit(1)                              # (not sys.exit: a rank blocked peers' collectives must not wait in atexit handlers)


if __name__ == "__main__":
    guarded_main()

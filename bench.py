#!/usr/bin/env python3
"""bench.py -- the headline measurement (BASELINE.json): PrimalCR++ pairwise-comparisons/sec
+ NDCG@10 on ml1m-shaped synthetic ratings, rank 100, lambda 5000, on N MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

A "step" is one outer iteration of pcrpp() (pcrpp.cpp:873-881): one truncated-Newton step on V
(gradient, <=10 CG Hessian-vector products, line search) and one Newton step per user on U -- the
same clock scope as the reference's "Iter k time" (no load, no init, no evaluation).  Inputs
(ratings, factors) are resident in HBM before the timed region starts.

value = #Omega * K / seconds, #Omega = #{(i,j,k): R_ij > R_ik} = the ordered pairs the objective
sums over.  N > 1 is WEAK scaling: each rank owns 6040 more users of the same item catalogue
(user-sharded; V-gradient and every Hessian-vector product are all-reduced over RCCL).

The JSON line also carries
  roofline      the kernel with the largest share of the timed region: algorithmic bytes per launch
                 / its average duration from HIP events recorded on the
                solver's stream during the timed region, against the 8 TB/s HBM3E peak
  cpu_baseline  the reference's own OpenMP path (oracle/_ref/omp-pmf-train, built from the unmodified
                reference) on the same data on this host's cores, 2 iterations (rank 0, N = 1 only);
                falls back to the single-thread C restatement on a user sample when _ref is absent.
"""
import argparse
import json
import os
import re
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)
TRAFFIC_FILE = "r02_traffic.json"   # PMC passes of this command at this round's kernels (tools/collect_profiles.sh)
USERS_PER_GPU = 6040
D2, NNZ_PER_GPU = 3952, 939809


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def algorithmic_bytes(slot, nnz_b, nu_b, d2, r, esz):
    """Compulsory HBM bytes of ONE launch of a per-user kernel over a length bin holding nnz_b
    ratings of nu_b users (ideal caching: every operand crosses HBM once).  DESIGN.md section 4."""
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
    """Does a rocprof kernel name belong to this HIP-event slot (class/workgroup size[.bound][g][c][t])?"""
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
    flags = tag.lstrip("0123456789.")
    block = tag[:len(tag) - len(flags)].partition(".")[0]
    big, clu = "g" in flags, "c" in flags
    if args[1] != block or (args[2] == "true") != big:
        return False
    if cls in ("vgrad", "vhv"):
        return (args[3] == "true") == (cls == "vhv")
    if cls == "ustep":
        return (args[3] != "1") == clu
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


def cpu_baseline(R, n_pairs, r, lam):
    """Reference OpenMP path on this host (kind "reference"), else the C restatement (kind "port").
    BASELINE.md 3.3: timed at -n <all cores of this box's share> AND at -n 1.
    Bounded: data sets beyond 2 M ratings are timed on a prefix of their users (same shape, fewer users)."""
    from oracle import oracle_py
    from primalcr_amd import synth
    cores = host_cores()
    sample_note = "the full data set"
    user, tuser = R.user, R.tuser
    if R.nnz > 2_000_000:
        nu = int(user[2_000_000])                 # whole users within the first 2 M ratings (triplets are user-sorted)
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
            d = synth.write_dir(R, os.path.join(td, "data"))
            it_all, it_one = 2, 1
            secs = ref_run(cores, it_all, d, td)
            secs1 = ref_run(1, it_one, d, td)
        return {"value": n_pairs * it_all / secs, "unit": "pairs/s", "cores": cores, "kind": "reference",
                "sample": f"omp-pmf-train -s 2 -k {r} -l {lam:g} -t {it_all} -p 0 -n {cores} on {sample_note}; "
                          f"'Iter {it_all} time' = {secs:.3f} s", "s_per_iter": secs / it_all,
                "single_thread": {"value": n_pairs * it_one / secs1, "unit": "pairs/s", "cores": 1, "s_per_iter": secs1 / it_one,
                                  "sample": f"the same command with -n 1 -t {it_one}: 'Iter {it_one} time' = {secs1:.3f} s"}}
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


def timed_run(pcr, torch, dist, ds, R, r, lam, prec, rank, N, local_rank, args, profile, shm_name, count_rows=False):
    """One solver, `warmup` untimed + exactly `steps` timed outer iterations (barrier + synchronize on both sides, MAX over
    ranks), then the quality after warmup + steps iterations."""
    p = pcr.Parameter(k=r, precision=prec, device=local_rank, do_predict=0, maxiter=1, **{"lambda": lam})
    with pcr.tuned(**({"count_rows": 1} if count_rows else {})):
        s = pcr.Solver(ds, p, rank, N)
    if N > 1:
        if args.comm == "p2p":
            s.comm_init_p2p(shm_name + ("_64" if prec == pcr.PCR_F64 else "_32") + ("c" if count_rows else ""))
        else:
            ids = [pcr.comm_unique_id() if rank == 0 else None]
            dist.broadcast_object_list(ids, src=0)
            s.comm_init(ids[0])
    s.set_factors(pcr.initial(R.d1, r), pcr.initial(R.d2, r))       # the reference's init stream (util.cpp:80)

    def barrier():
        s.sync()
        torch.cuda.synchronize()
        if N > 1:
            dist.barrier()

    objs = [rec["obj"] for rec in s.iterate(args.warmup)]
    prof_period = 0
    if profile:
        # sampled: every n-th launch of each kernel carries an event pair (an event pair costs ~3 us of queue time: every 4th
        # launch adds 6.6 % to the timed region, every 16th 1.9 %, with the same per-kernel averages)
        prof_period = args.profile_period if args.profile_period > 0 else max(1, min(16, 11 * args.steps // 12))
        s.profile(True, period=prof_period)
        s.profile_reset()
    barrier()
    rows0 = s.counter("ustep_row_gathers")
    t0 = time.perf_counter()
    inner = {"cg_v": 0, "ls_v": 0, "cg_u": 0, "ls_u": 0}
    # exactly K steps = K outer iterations (V step + U step) of the training loop, as pcr_train runs them (pcr_iterate)
    for rec in s.iterate(args.steps):
        objs.append(rec["obj"])
        for key in inner:
            inner[key] += rec[key]
    barrier()
    secs = time.perf_counter() - t0
    if N > 1:
        tt = torch.tensor([secs], device="cuda" if args.rendezvous == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        secs = float(tt.item())
    prof = s.profile_all() if profile else {}
    launches = {name: s.profile_launches(name) for name in prof}
    scope = {name: s.profile_scope(name) for name in prof}
    s.profile(False)
    te_err, te_ndcg = s.evaluate(1, 10)
    tr_err, tr_ndcg = s.evaluate(0, 10)
    out = dict(secs=secs, objs=objs, inner=inner, prof=prof, launches=launches, scope=scope, prof_period=prof_period,
               u_rows=s.counter("ustep_row_gathers") - rows0,
               te=(te_err, te_ndcg), tr=(tr_err, tr_ndcg), comm_nranks=s.comm_nranks(), shard=(s.first_user, s.n_users, s.nnz_local))
    s.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--shape", choices=["ml1m", "netflix", "yahoo"], default="ml1m",
                    help="ml1m: configs[1], 6040 users per GPU (weak scaling); netflix: configs[3], 480189 x 17770, 100 M ratings in "
                         "total, user-sharded over the GPUs (strong scaling); yahoo: configs[4] shape -- give --users (a user prefix)")
    ap.add_argument("--users", type=int, default=None, help="override the user count of the shape (total for netflix/yahoo, per GPU for ml1m)")
    ap.add_argument("--nnz", type=int, default=None, help="override the rating count likewise")
    ap.add_argument("--rank-k", type=int, default=None, help="factor rank (BASELINE: 100; yahoo: 200)")
    ap.add_argument("--lam", type=float, default=5000.0)
    ap.add_argument("--precision", choices=["f32", "f64"], default="f32")
    ap.add_argument("--comm", choices=["rccl", "p2p"], default="rccl", help="N > 1: ncclAllReduce, or the direct peer-to-peer exchange")
    ap.add_argument("--devices", default=None, help="HIP device of every local rank, e.g. 0,0 (rehearsal of the N > 1 path on a box with "
                                                    "fewer GPUs than ranks; needs --comm p2p and --rendezvous gloo: RCCL refuses two ranks on one device)")
    ap.add_argument("--rendezvous", choices=["nccl", "gloo"], default="nccl", help="torch.distributed backend of the barriers and the id broadcast")
    ap.add_argument("--tune", action="append", default=[], help="key=value launch knob (pcr_tune), repeatable -- for A/B runs")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-f64", action="store_true", help="skip the second timed run in the reference's arithmetic type")
    ap.add_argument("--no-profile", action="store_true", help="do not record per-kernel HIP events in the timed region")
    ap.add_argument("--profile-period", type=int, default=0,
                    help="event-time every n-th launch of each kernel (the first one included); 0 = as sparse as leaves a dozen "
                         "samples of the most frequent kernel (11 launches per step): min(16, 11 * steps / 12)")
    ap.add_argument("--verbose", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    N = args.gpus
    if world != N:
        if world == 1 and N > 1:
            raise SystemExit("bench.py --gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)")
        N = world
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

    if local_rank == 0 and not os.path.exists(os.path.join(os.path.dirname(os.path.abspath(__file__)), "primalcr_amd", "lib", "libprimalcr.so")):
        import __graft_entry__                   # clean checkout: compile the product first (no fallback exists)
        __graft_entry__.build()
    if N > 1:
        dist.barrier()                           # (every rank, whether or not it saw the library missing)
    import primalcr_amd as pcr
    from primalcr_amd import synth

    for kv in args.tune:
        pcr.tune(*kv.split("=", 1))
    r, lam = args.rank_k or (200 if args.shape == "yahoo" else 100), args.lam
    t0 = time.time()
    if args.shape == "ml1m":
        R = synth.generate("ml1m", d1=(args.users or USERS_PER_GPU) * N, nnz=(args.nnz or NNZ_PER_GPU) * N)
        scaling, shape_note = "weak", f"ml1m-shaped PrimalCR++ -k {r} -l {lam:g} (configs[1])"
    elif args.shape == "netflix":
        R = synth.generate_fast("netflix", d1=args.users, nnz=args.nnz)
        scaling, shape_note = "strong", f"Netflix-shaped PrimalCR++ -k {r} -l {lam:g} (configs[3])"
    else:
        R = synth.generate_fast("yahoo", users=(0, args.users or 225000))
        scaling, shape_note = "strong", f"Yahoo!Music-shaped PrimalCR++ -k {r} -l {lam:g} (configs[4]: first {R.d1} of 1.8 M users)"
    ds = pcr.Dataset.from_ratings(R)
    n_pairs = ds.count_pairs()
    if rank == 0:
        log(f"[data] {args.shape}-shaped x{N}: {R.d1} users x {R.d2} items, {R.nnz} ratings, {n_pairs} ordered pairs, "
            f"{len(R.tval)} test ratings ({time.time() - t0:.1f}s)")
    prec = pcr.PCR_F32 if args.precision == "f32" else pcr.PCR_F64
    shm = [f"/pcr_bench_{os.getpid()}_{int(time.time()) % 100000}" if rank == 0 else None]
    if N > 1:
        dist.broadcast_object_list(shm, src=0)
    run = timed_run(pcr, torch, dist, ds, R, r, lam, prec, rank, N, device, args, not args.no_profile, shm[0])
    # the same workload in the reference's arithmetic type (fp64 storage as well as fp64 accumulation), timed the same way
    run64 = None
    if prec == pcr.PCR_F32 and not args.no_f64:
        run64 = timed_run(pcr, torch, dist, ds, R, r, lam, pcr.PCR_F64, rank, N, device, args, False, shm[0])
    # diagnostic replay of the same iterations with the U-step kernels counting the rows of V they gather (an extra atomic per
    # user, so it is kept out of the timed run): the rating-weighted pass count of the U step
    run["u_rows"] = timed_run(pcr, torch, dist, ds, R, r, lam, prec, rank, N, device, args, False, shm[0], count_rows=True)["u_rows"]
    secs, objs, inner, prof = run["secs"], run["objs"], run["inner"], run["prof"]
    te_err, te_ndcg = run["te"]; tr_err, tr_ndcg = run["tr"]
    prof_period = run["prof_period"]

    if rank != 0:
        if N > 1:
            dist.barrier(); dist.destroy_process_group()
        return

    # ---- per-kernel rooflines (rank 0's shard)
    roof, roof_phase, kernels = None, {}, {}
    esz = 4 if prec == pcr.PCR_F32 else 8
    if prof:
        traffic, traffic_src = {}, None
        tpath = os.path.join(ROOT, "profiles", TRAFFIC_FILE)      # PMC passes of this command (tools/pmc_traffic.py)
        if os.path.exists(tpath) and args.shape == "ml1m" and not args.users and not args.nnz and r == 100 and N == 1 and prec == pcr.PCR_F32:
            tj = json.load(open(tpath))             # the PMC passes were taken on exactly this workload
            traffic, traffic_src = tj.get("kernels", tj), tj.get("source")
        # Launches are SAMPLED, so a slot's time in the region is its average times ALL its launches.
        est = {name: ((ms / n) * max(run["launches"][name], n) if n else 0.0) for name, (ms, n) in prof.items()}
        total_ms = sum(v for k, v in est.items() if not k.startswith("wall:"))
        for name, (ms, n) in prof.items():
            cls, _, tag = name.partition("/")
            if cls not in SLOT_KERNEL or n == 0:
                continue
            nnz_b, nu_b = run["scope"][name]              # ratings / users one launch of this slot covers
            ab = algorithmic_bytes(name, nnz_b, nu_b, R.d2, r, esz)
            avg_s = ms / n / 1e3
            tr = None
            # several slots can share one kernel symbol (k_ustep of one workgroup size serves several length classes):
            # a per-symbol PMC average cannot be split between them, so those slots carry no traffic figure
            shared = sum(1 for other in prof if other != name and prof[other][1] and other.partition("/")[0] == cls and
                         any(slot_kernel_match(other, kn, args.precision) and slot_kernel_match(name, kn, args.precision) for kn in traffic))
            for kname, t in traffic.items():
                if not shared and slot_kernel_match(name, kname, args.precision):
                    # MI355X_MICROARCH.md (HBM): FETCH_SIZE counts wide coalesced reads at half their bytes on gfx950 ->
                    # doubled; WRITE_SIZE is exact
                    tr = int(2 * t["fetch_bytes_per_launch_raw"] + t["write_bytes_per_launch"])
            kernels[name] = {"avg_us": round(avg_s * 1e6, 2), "timed_launches": int(n), "launches": int(run["launches"][name]),
                             "gpu_time_share": round(est[name] / total_ms, 4), "concurrent_group": cls in ("ustep", "prepare", "eval"),
                             "algorithmic_bytes": int(ab), "achieved_GBs": round(ab / avg_s / 1e9, 2),
                             "frac_hbm_peak": round(ab / avg_s / 1e9 / HBM_PEAK_GBS, 5), "traffic_bytes": tr}
        if args.verbose:
            for k, (ms, n) in sorted(prof.items(), key=lambda kv: -est[kv[0]]):
                extra = f"  alg {kernels[k]['achieved_GBs']:8.1f} GB/s  gpu-time share {100 * kernels[k]['gpu_time_share']:5.1f} %" if k in kernels else ""
                log(f"  {k:14s} {est[k]:9.3f} ms  {n:6d} timed  {1e3 * ms / max(n, 1):9.1f} us/launch{extra}")
        # dominant kernel = the slot with the most GPU time (average duration x launches), the way `rocprofv3 --stats` ranks
        # kernels -- concurrent length classes are NOT discounted for running side by side
        dom = max(kernels, key=lambda k: kernels[k]["gpu_time_share"])
        kd = kernels[dom]
        roof = {"bound": "hbm", "kernel": dom, "achieved": kd["achieved_GBs"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": kd["frac_hbm_peak"], "traffic": kd["traffic_bytes"], "avg_launch_us": kd["avg_us"],
                "launches_timed": kd["timed_launches"], "algorithmic_bytes_per_launch": kd["algorithmic_bytes"],
                "share_of_gpu_time": kd["gpu_time_share"],
                "traffic_source": (traffic_src or f"profiles/{TRAFFIC_FILE}") + ": stored rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this "
                                  "command (separate runs), per launch, 2 x FETCH_SIZE + WRITE_SIZE (MI355X_MICROARCH.md); not "
                                  "measured by this run" if kd["traffic_bytes"] is not None else None,
                "note": "dominant = largest GPU time (average launch duration x launches, HIP events on the launch stream), as "
                        f"rocprofv3 --stats ranks kernels; every {prof_period}th launch of a kernel is event-timed (once-per-step "
                        "kernels every 4th); all slots in 'kernels', phases in 'roofline_phase' (DESIGN.md 3.5, 4)"}
        # phases: algorithmic bytes of everything a phase launches per step / its wall time per step
        def phase(names, wall_ms_per_step):
            ab = sum(kernels[k]["algorithmic_bytes"] * run["launches"][k] / args.steps for k in names)
            return {"bound": "hbm", "algorithmic_bytes_per_step": int(ab), "wall_us_per_step": round(1e3 * wall_ms_per_step, 1),
                    "achieved": round(ab / (wall_ms_per_step / 1e3) / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(ab / (wall_ms_per_step / 1e3) / 1e9 / HBM_PEAK_GBS, 5),
                    "share_of_step": round(wall_ms_per_step / (1e3 * secs / args.steps), 4)}
        un = [k for k in kernels if k.startswith("ustep/")]
        if un and prof.get("wall:ustep", (0, 0))[1]:
            wm, wn = prof["wall:ustep"]
            u_gather = run["u_rows"] / args.steps / N * r * esz           # this rank's share (the counter is the all-rank total)
            roof_phase["u_step"] = dict(phase(un, wm / wn), kernels=un, gathered_row_bytes_per_step=int(u_gather),
                                        gather_GBs=round(u_gather / (wm / wn / 1e3) / 1e9, 1),
                                        note="all length classes of k_ustep, launched side by side: sum of their algorithmic bytes / fork..join wall "
                                             "time on the solver's stream; gather_GBs = rows of V actually gathered (counted in the kernel: per user "
                                             "1 + 2 per CG iteration + 1 per line-search try, x its ratings) x row bytes / that wall time -- the "
                                             "L2 -> CU row-gather rate this phase runs at (measured ceiling 16.8-18.8 TB/s, MI355X_MICROARCH.md)")
        vn = [k for k in kernels if k.partition("/")[0] in ("sddmm", "spmm", "spmm_fin", "vhv", "vgrad", "cg", "prepare")]
        if vn:
            u_wall = roof_phase.get("u_step", {}).get("wall_us_per_step", 0.0) / 1e3
            v_ms = 1e3 * secs / args.steps - u_wall               # the two half steps alternate on one stream: the rest of a step is the V step
            v_gather = ((n_ls_v := inner["ls_v"] / args.steps) + 2 * (inner["cg_v"] / args.steps) + 1) * esz * r * run["shard"][2]
            roof_phase["v_step"] = dict(phase(vn, v_ms), kernels=vn, gathered_row_bytes_per_step=int(v_gather),
                                        gather_GBs=round(v_gather / (v_ms / 1e3) / 1e9, 1),
                                        note="gradient + CG (SDDMM, sweep, SpMM, finish, vector update) + line search, back to back on the "
                                             "solver's stream: step time minus the U step's wall time; gather_GBs = (1 SpMM + n_cg x (SDDMM + "
                                             "SpMM) + n_ls SDDMM) x ratings x row bytes / that time")
    cpu = None
    if N == 1 and not args.no_cpu:
        cpu = cpu_baseline(R, n_pairs, r, lam)

    value = n_pairs * args.steps / secs
    passes = (1 + (inner["cg_v"] + inner["ls_v"]) / args.steps) + (1 + (inner["cg_u"] + inner["ls_u"]) / args.steps / max(R.d1, 1))
    # SURVEY 8d, the whole-iteration figure: compulsory bytes W of one outer iteration with ideal caching (esz-byte factors,
    # int32 item, uint8 level, esz-byte m, uint32 permutation) at the EXECUTED inner counts, over the measured time per
    # iteration -- all ranks' bytes over the job's time, against N x 8 TB/s.
    esz_w = esz
    n_cg, n_ls = inner["cg_v"] / args.steps, inner["ls_v"] / args.steps
    B_csr, F_U, F_V = 5 * R.nnz + 8 * (R.d1 + 1), esz_w * r * R.d1, esz_w * r * R.d2 * N      # V is replicated on every rank
    P_m, P_sort = B_csr + F_U + F_V + esz_w * R.nnz, 8 * R.nnz
    P_hv, P_obj, P_u = B_csr + 8 * R.nnz + F_U + 2 * F_V, B_csr + 8 * R.nnz, B_csr + esz_w * R.nnz + 2 * F_U + F_V
    W = P_m + P_sort + P_hv + n_cg * P_hv + P_obj + n_ls * (P_m + P_sort + P_obj) + P_u
    it_roof = {"bound": "hbm", "algorithmic_bytes_per_iteration": int(W), "achieved": round(W / (secs / args.steps) / 1e9, 2),
               "peak": HBM_PEAK_GBS * N, "unit": "GB/s", "frac": round(W / (secs / args.steps) / 1e9 / (HBM_PEAK_GBS * N), 5),
               "note": "SURVEY 8d: W = P_m + P_sort + P_g + n_cg P_Hv + P_obj + n_ls (P_m + P_sort + P_obj) + P_U at the executed "
                       "n_cg, n_ls; the factor tables of this shape are L2-resident, so the path is gather/latency-bound and "
                       "this fraction is small by construction (DESIGN.md 3.5)"}
    # SURVEY 8d, secondary (diagnostic) figure: row-gather bytes.  One SDDMM or SpMM half-pass moves G = esz * r bytes per
    # rating; per outer iteration the V side makes (1 + n_ls) SDDMMs of the prepares + n_cg of the CG + (1 + n_cg) SpMMs, the
    # U side per rating 1 (gradient) + 2 per CG iteration + 1 per line-search try.
    G = esz_w * r * R.nnz
    n_cg_u, n_ls_u = inner["cg_u"] / args.steps / max(R.d1, 1), inner["ls_u"] / args.steps / max(R.d1, 1)
    u_half_passes = run["u_rows"] / args.steps / max(R.nnz, 1)       # counted in k_ustep: rating-weighted, not user-averaged
    gather_passes = (n_ls + n_cg) + (1 + n_cg) + u_half_passes
    gather = {"bytes_per_half_pass": int(G), "half_passes_per_iteration": round(gather_passes, 2),
              "u_side_half_passes": round(u_half_passes, 2), "u_side_user_average": round(1 + 2 * n_cg_u + n_ls_u, 2),
              "achieved_GBs": round(gather_passes * G / (secs / args.steps) / 1e9, 1),
              "note": "row gathers (one esz*r-byte factor row per rating and half-pass) sustained over the WHOLE iteration, all "
                      "ranks; on this shape they are served by the L2s (U side: rows counted by the kernel -- long users run more CG "
                      "iterations than the user average, so the rating-weighted pass count is the higher one)"}
    out = {
        "metric": "pairwise-comparisons/sec", "value": value, "unit": "pairs/s", "n_gpus": N, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": 1e3 * secs / args.steps, "higher_is_better": True, "scaling": scaling,
        "vs_baseline": None, "dtype": "f32" if prec == pcr.PCR_F32 else "f64", "data": "synthetic",
        "config": {"workload": f"{shape_note}; {R.d1} users x {R.d2} items, "
                               f"{R.nnz} ratings, {n_pairs} ordered pairs; 1 step = 1 outer iteration (V step + U step)",
                   "solver": "PrimalCR++", "rank": r, "lambda": lam, "parallelism": f"user-sharded x{N}",
                   "accumulation": "f64", "storage": "f32" if prec == pcr.PCR_F32 else "f64",
                   "exchange": None if N == 1 else args.comm},
        "ndcg10_test": te_ndcg, "pairwise_error_test": te_err, "ndcg10_train": tr_ndcg, "pairwise_error_train": tr_err,
        "outer_iterations_run": args.warmup + args.steps, "objective": objs[-1],
        "inner_per_step": {k: v / args.steps for k, v in inner.items()},
        # SURVEY 8d, kernel-level figure: ordered pairs swept per second over the EXECUTED sweep passes of a step
        # (V side: gradient + Hessian-vector products + line-search objectives; U side the same per user, averaged)
        "passes_per_step": passes, "sweep_pairs_per_s": value * passes, "s_per_iter": secs / args.steps,
        "comm_nranks": run["comm_nranks"],
        "roofline": roof, "roofline_phase": roof_phase, "roofline_iteration": it_roof, "gather": gather, "cpu_baseline": cpu,
        "kernels": kernels,
    }
    if run64:
        # the reference computes in fp64 throughout: the same K steps with fp64 storage, same clock, same barriers
        out["f64"] = {"dtype": "f64", "ms_per_step": 1e3 * run64["secs"] / args.steps, "value": n_pairs * args.steps / run64["secs"],
                      "unit": "pairs/s", "ndcg10_test": run64["te"][1], "pairwise_error_test": run64["te"][0],
                      "objective": run64["objs"][-1], "inner_per_step": {k: v / args.steps for k, v in run64["inner"].items()},
                      "note": "second timed run of the same workload with U, V, m and the CG vectors stored in fp64 (the reference's "
                              "arithmetic type); 'value' above is the fp32-storage / fp64-accumulation run the north star allows "
                              "('within fp32 tolerance')"}
        out["f64_minus_f32"] = {"ndcg10_test": run64["te"][1] - te_ndcg, "pairwise_error_test": run64["te"][0] - te_err,
                                "objective_rel": run64["objs"][-1] / objs[-1] - 1}
    if cpu:
        out["speedup_vs_cpu_baseline"] = value / cpu["value"]
    print(json.dumps(out), flush=True)
    if N > 1:
        dist.barrier(); dist.destroy_process_group()


if __name__ == "__main__":
    main()

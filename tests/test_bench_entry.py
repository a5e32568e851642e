"""bench.py as the driver starts it.  `python bench.py --gpus N` with N > 1 must start its own ranks (child processes of a
parent that never touches a GPU) and print rank 0's JSON line; the N = 2 path -- per-rank shard generation,
pcr_solver_create_shard, every exchange step, the max-over-ranks clock, the line -- is rehearsed with two ranks on the one
GPU through the peer-to-peer exchange (RCCL refuses two ranks on one device) and compared with ONE rank training the same
2 x 2000 users."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import primalcr_amd as pcr
from primalcr_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def run_bench(*args, timeout=600):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return subprocess.run([sys.executable, BENCH, *args], capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)


def test_initial_rows_are_the_rows_of_initial():
    full = pcr.initial(37, 5)
    assert np.array_equal(pcr.initial_rows(37, 5, 0, 37), full)
    assert np.array_equal(pcr.initial_rows(37, 5, 11, 9), full[11:20])
    assert pcr.initial_rows(37, 5, 37, 0).shape == (0, 5)
    with pytest.raises(pcr.PcrError):
        pcr.initial_rows(37, 5, 30, 8)


def test_user_blocks_of_one_item_catalogue():
    """Rank q's block of the weak-scaling workload: its own users (seed + q), the item ground truth of seed 0's set."""
    a = synth.generate("tiny", seed=5)
    b = synth.generate("tiny", seed=6, item_seed=5)
    c = synth.generate("tiny", seed=6)
    assert np.array_equal(synth.generate("tiny", seed=5, item_seed=5).val, a.val)
    assert np.array_equal(b.item, c.item) and np.array_equal(b.user, c.user)          # the same users and item sets ...
    assert not np.array_equal(b.val, c.val)                                            # ... rated against another catalogue


def test_yahoo_shape_partitions_into_eight_balanced_shards_without_generating_it():
    """configs[4] on 8 GPUs: every rank needs only the per-user counts (1.8 M integers) to find its nnz-balanced user range."""
    ctr, cte = synth.generate_fast("yahoo", counts_only=True)
    assert ctr.shape == (1_800_000,) and int(ctr.sum()) == 700_000_000
    index = np.concatenate([[0], np.cumsum(ctr)])
    b = pcr.partition_users(index, 8)
    nnz = np.diff(index[b])
    assert b[0] == 0 and b[-1] == 1_800_000 and (np.diff(b) > 0).all()
    assert np.abs(nnz / 87_500_000 - 1).max() < 0.002, nnz


@pytest.mark.timeout(300)
def test_gpus_flag_starts_its_own_ranks_and_fails_loudly_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible: the N = 2 run itself is test_two_ranks_on_one_gpu_equal_one_rank")
    p = run_bench("--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu")
    assert p.returncode != 0
    assert "launching 2 ranks" in p.stderr and "no GPU visible" in p.stderr and "must be launched with" not in p.stderr
    assert not p.stdout.strip()                                                    # no JSON line from a failed job


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_two_ranks_on_one_gpu_equal_one_rank():
    users, nnz, r, steps, warmup = 2000, 250000, 100, 2, 1
    p = run_bench("--gpus", "2", "--comm", "p2p", "--devices", "0,0", "--rendezvous", "gloo", "--steps", str(steps), "--warmup",
                  str(warmup), "--users", str(users), "--nnz", str(nnz), "--no-cpu", "--no-f64")
    assert p.returncode == 0, p.stderr[-3000:]
    line = json.loads(p.stdout.strip().split("\n")[-1])
    assert line["n_gpus"] == 2 and line["comm_nranks"] == 2 and line["scaling"] == "weak" and line["config"]["exchange"] == "p2p"
    assert line["steps"] == steps and line["warmup"] == warmup
    assert line["shards"] == [[0, users, nnz], [users, users, nnz]]
    assert line["roofline"] and line["roofline"]["binding"]["level"] == "l2-gather"
    # the same 2 x 2000 users in ONE solver
    blocks = [synth.generate("ml1m", seed=synth.SEED + q, d1=users, nnz=nnz, item_seed=synth.SEED if q else None) for q in range(2)]
    cat = lambda f: np.concatenate([getattr(blocks[0], f), getattr(blocks[1], f)])
    ds = pcr.Dataset.from_triplets(2 * users, blocks[0].d2, np.concatenate([blocks[0].user, blocks[1].user + users]), cat("item"), cat("val"),
                                   np.concatenate([blocks[0].tuser, blocks[1].tuser + users]), cat("titem"), cat("tval"))
    assert ds.count_pairs() * steps / (line["ms_per_step"] * 1e-3 * steps) == pytest.approx(line["value"], rel=1e-9)
    s = pcr.Solver(ds, pcr.Parameter(k=r, precision=pcr.PCR_F32, do_predict=0, **{"lambda": 5000.0}))
    s.set_factors(pcr.initial(2 * users, r), pcr.initial(blocks[0].d2, r))
    recs = s.iterate(warmup + steps)
    err, ndcg = s.evaluate(1, 10)
    assert recs[-1]["obj"] == pytest.approx(line["objective"], rel=2e-5)             # fp32 storage: summation order differs
    assert ndcg == pytest.approx(line["ndcg10_test"], abs=2e-4) and err == pytest.approx(line["pairwise_error_test"], abs=2e-4)
    assert line["inner_per_step"]["cg_v"] == sum(x["cg_v"] for x in recs[warmup:]) / steps


def _two_devices():
    try:
        import torch
        return torch.cuda.device_count() >= 2
    except Exception:
        return False


@pytest.mark.gpu
@pytest.mark.timeout(900)
@pytest.mark.skipif(not _two_devices(), reason="needs two physical GPUs (the driver's multi-GPU node)")
@pytest.mark.parametrize("comm,chunks", [("rccl", 1), ("rccl", 3), ("p2p", 1), ("p2p", 3)])
def test_two_physical_gpus_equal_one_rank(comm, chunks):
    """RCCL with more than one rank, the peer-to-peer exchange across xGMI and the item-range overlap have only ever run
    where a second GPU exists: when there is one, hold them to the one-rank trajectory."""
    users, nnz, steps, warmup = 2000, 250000, 2, 1
    common = ["--steps", str(steps), "--warmup", str(warmup), "--users", str(users), "--nnz", str(nnz), "--no-cpu", "--no-f64", "--precision", "f64",
              "--no-profile"]
    two = run_bench("--gpus", "2", "--comm", comm, "--tune", f"allreduce_chunks={chunks}", *common)
    assert two.returncode == 0, two.stderr[-3000:]
    a = json.loads(two.stdout.strip().split("\n")[-1])
    assert a["n_gpus"] == 2 and a["comm_nranks"] == 2
    blocks = [synth.generate("ml1m", seed=synth.SEED + q, d1=users, nnz=nnz, item_seed=synth.SEED if q else None) for q in range(2)]
    cat = lambda f: np.concatenate([getattr(blocks[0], f), getattr(blocks[1], f)])
    ds = pcr.Dataset.from_triplets(2 * users, blocks[0].d2, np.concatenate([blocks[0].user, blocks[1].user + users]), cat("item"), cat("val"),
                                   np.concatenate([blocks[0].tuser, blocks[1].tuser + users]), cat("titem"), cat("tval"))
    s = pcr.Solver(ds, pcr.Parameter(k=100, precision=pcr.PCR_F64, do_predict=0, **{"lambda": 5000.0}))
    s.set_factors(pcr.initial(2 * users, 100), pcr.initial(blocks[0].d2, 100))
    recs = s.iterate(warmup + steps)
    assert recs[-1]["obj"] == pytest.approx(a["objective"], rel=1e-10)
    assert s.evaluate(1, 10)[1] == pytest.approx(a["ndcg10_test"], abs=1e-9)


def test_every_ustep_class_is_one_kernel_symbol_in_the_committed_profiles():
    """bench.py prices a HIP-event slot with the PMC bytes of 'its' kernel symbol (profiles/r03_traffic.json): every U-step length
    class of the committed bench line (ml1m and the Netflix-shaped sub-record) must match exactly one k_ustep symbol of the
    rocprofv3 kernel stats of the same shape, and no symbol may serve two classes."""
    import csv
    sys.path.insert(0, ROOT)
    import bench
    line = json.loads(open(os.path.join(ROOT, "profiles", "r03_d_bench.json")).read().strip().split("\n")[-1])
    for shape, rec in (("ml1m", line), ("netflix", line["netflix"])):
        names = [r["Name"] for r in csv.DictReader(open(os.path.join(ROOT, "profiles", f"r03_d_{shape}_f32_kernel_stats.csv")))
                 if r["Name"].startswith("void k_ustep<float")]
        slots = [k for k in rec["kernels"] if k.startswith("ustep/")]
        assert len(slots) >= 6 and len(names) >= len(slots)
        owner = {}
        for sl in slots:
            hit = [n for n in names if bench.slot_kernel_match(sl, n, "f32")]
            assert len(hit) == 1, (shape, sl, hit)
            assert hit[0] not in owner, (shape, sl, owner[hit[0]])
            owner[hit[0]] = sl
        dom = rec["roofline"]["kernel"]
        if dom.startswith("ustep/"):
            assert rec["roofline"]["traffic"] and rec["roofline"]["traffic_over_algorithmic"] > 1, (shape, rec["roofline"])

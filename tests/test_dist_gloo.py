"""Multi-process CPU tests of the N > 1 path (gloo, world_size 2 and 8, 127.0.0.1).

No GPU here, and the product has no CPU compute path, so the *sharded algorithm* is exercised with
the oracle standing in for the per-shard kernels (tests may use the oracle as a checker): each rank
owns the user range `pcr_partition_users` gives it, computes its partial of the V-gradient /
Hessian-vector product on that range only (rank 0 carries the lambda term -- exactly how
pcr_solver.hip seeds the buffers), all-reduces over gloo, and runs the replicated CG recurrence of
solve_delta_new.  The combined results must equal the single-process oracle.  Also covered: the
128-byte communicator id broadcast that bench.py uses for the RCCL bootstrap.
"""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist

    import primalcr_amd as pcr
    from oracle.oracle_py import CSR, Oracle
    from primalcr_amd import synth

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        orc = Oracle()
        R = synth.generate("small", seed=4, d1=120, d2=70, nnz=3000, mu=3.0, sigma=0.8)
        r, lam = 6, 20.0
        X = orc.build_csr(R.d1, R.d2, R.user, R.item, R.val)
        U, V = orc.initial(R.d1, r) * 0.5, orc.initial(R.d2, r) * 0.5
        # --- the product's partitioner decides the shard
        b = pcr.partition_users(X.idx, world)
        u0, u1 = int(b[rank]), int(b[rank + 1])
        z0, z1 = int(X.idx[u0]), int(X.idx[u1])
        Xs = CSR(u1 - u0, X.d2, X.idx[u0:u1 + 1] - z0, X.item[z0:z1], X.val[z0:z1])
        Us = U[u0:u1]
        lam_r = lam if rank == 0 else 0.0            # rank 0 seeds lambda*V / lambda*p, the others 0

        def allreduce(a):
            t = torch.from_numpy(np.ascontiguousarray(a))
            dist.all_reduce(t)
            return t.numpy()

        m_s = orc.comp_m(Us, V, Xs)
        g = allreduce(orc.obtain_g_new(Us, V, Xs, m_s, lam_r))
        loss = allreduce(np.array([orc.objective_new(m_s, Us * 0, V * 0, Xs, 0.0)]))[0]
        obj = loss + lam * ((U ** 2).sum() + (V ** 2).sum()) / 2.0

        def hv(p):
            return allreduce(orc.compute_Ha_new(p, m_s, Us, Xs, lam_r))

        # replicated CG recurrence (pcrpp.cpp:335-358): identical scalars on every rank, no communication
        delta = np.zeros_like(g); rr = -g; p = g.copy()
        err = np.sqrt((rr ** 2).sum()) * 0.01
        its = 0
        for _ in range(10):
            Hp = hv(p); its += 1
            pHp = (p * Hp).sum()
            alpha = -(rr * p).sum() / pHp
            delta = delta + alpha * p
            rr = rr + alpha * Hp
            if np.sqrt((rr ** 2).sum()) < err:
                break
            p = -rr + ((rr * Hp).sum() / pHp) * p
        # --- communicator-id bootstrap exactly as bench.py does it
        ids = [bytes(range(128)) if rank == 0 else None]
        dist.broadcast_object_list(ids, src=0)
        assert ids[0] == bytes(range(128))
        its_all = [None] * world
        dist.all_gather_object(its_all, its)
        assert len(set(its_all)) == 1, "ranks disagree on the CG iteration count"
        # U step is embarrassingly parallel: each rank updates its own rows
        m_full = orc.comp_m(U, V, X)
        Un_s, obj_s, _ = orc.update_U_new(Xs, m_s, lam, 1.0, V, Us)
        obj_u = allreduce(np.array([obj_s - lam / 2.0 * (V ** 2).sum()]))[0] + lam / 2.0 * (V ** 2).sum()
        gathered = [None] * world
        dist.all_gather_object(gathered, (u0, Un_s))
        if rank == 0:
            g_ref = orc.obtain_g_new(U, V, X, m_full, lam)
            d_ref, its_ref = orc.solve_delta_new(g_ref, m_full, U, X, lam)
            obj_ref = orc.objective_new(m_full, U, V, X, lam)
            Un_ref, obju_ref, _ = orc.update_U_new(X, m_full, lam, 1.0, V, U)
            Un = np.concatenate([x[1] for x in sorted(gathered, key=lambda t: t[0])])
            np.savez(os.path.join(out_dir, "res.npz"), g=np.abs(g - g_ref).max() / np.abs(g_ref).max(),
                     d=np.abs(delta - d_ref).max() / np.abs(d_ref).max(), its=its - its_ref,
                     obj=abs(obj / obj_ref - 1), U=np.abs(Un - Un_ref).max(), obju=abs(obj_u / obju_ref - 1))
    finally:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world", [2, 8])
def test_sharded_v_step_and_u_step(world, tmp_path):
    """world 8 = the target machine's rank count (no GPU box lets eight processes share its one card, NOTES.md round 5: the 8-way
    partition, per-rank partials with rank 0 carrying the lambda term, the replicated CG recurrence and the gather of U rows run
    here over gloo)."""
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    res = np.load(tmp_path / "res.npz")
    assert res["g"] < 1e-12 and res["d"] < 1e-9 and res["its"] == 0
    assert res["obj"] < 1e-12 and res["U"] < 1e-12 and res["obju"] < 1e-12

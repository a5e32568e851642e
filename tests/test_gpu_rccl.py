"""GPU test of the RCCL plumbing on ONE GPU: a 1-rank communicator is created through the same
bootstrap bench.py uses (pcr_comm_unique_id -> pcr_solver_comm_init) and every collective of the
training loop then really goes through ncclAllReduce on the solver's stream.  Results must equal
the communicator-free run bit for bit.  torch is imported first, as in bench.py, so the test also
covers libprimalcr's RCCL/HIP runtime coexisting with PyTorch's in one process."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_training_through_rccl_single_rank():
    import torch  # noqa: F401  (library coexistence)
    import primalcr_amd as pcr
    from primalcr_amd import synth
    R = synth.generate("small", seed=2)
    ds = pcr.Dataset.from_ratings(R)
    U0, V0 = pcr.initial(R.d1, 16), pcr.initial(R.d2, 16)
    out = []
    for use_comm in (False, True):
        s = pcr.Solver(ds, pcr.Parameter(k=16, maxiter=2, do_predict=1, **{"lambda": 100.0}))
        if use_comm:
            s.comm_init(pcr.comm_unique_id())
        s.set_factors(U0, V0)
        recs, _ = s.train()
        U, V = s.get_factors()
        out.append((recs, U, V))
        s.close()
    (r0, U_a, V_a), (r1, U_b, V_b) = out
    assert np.array_equal(U_a, U_b) and np.array_equal(V_a, V_b)
    for a, b in zip(r0, r1):
        assert a["obj"] == b["obj"] and a["test_ndcg"] == b["test_ndcg"] and a["cg_v"] == b["cg_v"]

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
        # the pipelined loop bench.py times (pcr_iterate), from the same start, then the evaluator
        s.set_factors(U0, V0)
        it = s.iterate(2)
        Ui, Vi = s.get_factors()
        out.append((recs, U, V, it, Ui, Vi, s.evaluate(1, 10)))
        s.close()
    (r0, U_a, V_a, i0, Ui_a, Vi_a, e0), (r1, U_b, V_b, i1, Ui_b, Vi_b, e1) = out
    assert np.array_equal(U_a, U_b) and np.array_equal(V_a, V_b)
    assert np.array_equal(Ui_a, Ui_b) and np.array_equal(Vi_a, Vi_b) and e0 == e1
    assert np.array_equal(Ui_a, U_a) and np.array_equal(Vi_a, V_a)      # pipelined == step by step
    for a, b in zip(r0, r1):
        assert a["obj"] == b["obj"] and a["test_ndcg"] == b["test_ndcg"] and a["cg_v"] == b["cg_v"]
    for a, b in zip(i0, i1):
        assert a["obj"] == b["obj"] and a["cg_v"] == b["cg_v"] and a["cg_u"] == b["cg_u"]


def test_item_range_overlap_through_rccl_single_rank():
    """The N > 1 overlap form under RCCL itself: with allreduce_chunks the SpMM runs item range by item range, range r's
    ncclAllReduce is queued on the exchange stream (ar_st) behind an event while range r + 1 computes on the solver's stream,
    and the objective's scalar all-reduce goes through the solver's stream in between (launch_spmm / allreduce_T /
    allreduce_f64).  A 1-rank communicator executes exactly that interleaving of streams, events and collectives; the
    trajectory must equal the one-all-reduce-per-vector run and the communicator-free run bit for bit.
    Replaces the omp atomics of pcrpp.cpp:240-243, :323-327."""
    import torch  # noqa: F401
    import primalcr_amd as pcr
    from primalcr_amd import synth
    R = synth.generate("small", seed=7)
    ds = pcr.Dataset.from_ratings(R)
    U0, V0 = pcr.initial(R.d1, 24), pcr.initial(R.d2, 24)
    runs = {}
    for name, knobs, use_comm in (("plain", {}, False), ("rccl", {}, True), ("rccl-ranges", {"allreduce_chunks": 4}, True),
                                  ("ranges-no-comm", {"allreduce_chunks": 4}, False)):
        for prec in (pcr.PCR_F64, pcr.PCR_F32):
            with pcr.tuned(**knobs):
                s = pcr.Solver(ds, pcr.Parameter(k=24, precision=prec, do_predict=0, **{"lambda": 50.0}))
            if use_comm:
                s.comm_init(pcr.comm_unique_id())
                assert s.comm_nranks() == 1
            s.set_factors(U0, V0)
            g = s.comp_m() is not None and s.obtain_g()              # one all-reduced vector read back directly
            it = s.iterate(3)
            U, V = s.get_factors()
            runs[name, prec] = (g, [(r["obj"], r["cg_v"], r["cg_u"], r["ls_u"]) for r in it], U, V, s.evaluate(1, 10))
            s.close()
    for prec in (pcr.PCR_F64, pcr.PCR_F32):
        rt, at = (1e-11, 1e-9) if prec == pcr.PCR_F64 else (2e-4, 1e-3)
        # the same kernels with and without RCCL in the way: item ranges switch the fused dot products off either way, so the
        # communicator-free run and the RCCL run execute the same arithmetic -> the same bits
        ga, ita, Ua, Va, ea = runs["ranges-no-comm", prec]
        gb, itb, Ub, Vb, eb = runs["rccl-ranges", prec]
        assert np.array_equal(ga, gb) and ita == itb and np.array_equal(Ua, Ub) and np.array_equal(Va, Vb) and ea == eb
        # against the plain run (dot products fused into k_spmm_fin: another summation order) and the one-all-reduce-per-vector
        # RCCL run: the same trajectory to rounding, identical inner-iteration counts
        for other in ("plain", "rccl"):
            g0, it0, U0_, V0_, e0 = runs[other, prec]
            assert np.allclose(ga, g0, rtol=rt, atol=at), (other, prec)
            assert [x[1:] for x in ita] == [x[1:] for x in it0], (other, prec)
            assert np.allclose([x[0] for x in ita], [x[0] for x in it0], rtol=rt)
            assert np.allclose(Ua, U0_, rtol=0, atol=rt * np.abs(U0_).max() * 100) and np.allclose(Va, V0_, rtol=0, atol=rt * np.abs(V0_).max() * 100)
            assert np.allclose(ea, e0, atol=rt * 10)


def test_headline_config_matches_reference_binary(tmp_path):
    """BASELINE configs[1]: ml1m-shaped synthetic, PrimalCR++ -k 100 -l 5000, 3 outer iterations, default
    (fp32) precision on the GPU vs the UNMODIFIED reference binary (oracle/_ref/omp-pmf-train, all host
    cores) on the same files: objective within 1e-3 relative, NDCG@10 / pairwise error within 1e-3."""
    import os
    import re
    import subprocess
    import primalcr_amd as pcr
    from oracle import oracle_py
    from primalcr_amd import synth
    if not os.path.exists(oracle_py.REF_TRAIN):
        pytest.skip("oracle/_ref was not built (no /root/reference at build time)")
    R = synth.generate("ml1m")
    d = synth.write_dir(R, str(tmp_path / "ml1m"))
    iters = 3
    out = subprocess.run([oracle_py.REF_TRAIN, "-k", "100", "-l", "5000", "-t", str(iters), "-n", "16",
                          d, str(tmp_path / "ref.model")], cwd=tmp_path, capture_output=True, text=True, check=True).stdout
    objs = [float(x) for x in re.findall(r"^Iter \d+ time \S+ obj (\S+)$", out, re.M)]
    te = [(float(a), float(b)) for a, b in re.findall(r"^\(Testing\) pairwise error is (\S+) and ndcg is (\S+)$", out, re.M)]
    tr = [(float(a), float(b)) for a, b in re.findall(r"^\(Training\) pairwise error is (\S+) and ndcg is (\S+)$", out, re.M)]
    ds = pcr.Dataset.load(d)
    s = pcr.Solver(ds, pcr.Parameter(k=100, maxiter=iters, **{"lambda": 5000.0}))
    s.set_factors(pcr.initial(R.d1, 100), pcr.initial(R.d2, 100))
    recs, _ = s.train()
    assert len(objs) == iters + 1
    for k, rec in enumerate(recs):
        # the reference's printed objective after a U step is racy for -n > 1 (shared obj_u_new, pcrpp.cpp:822-832):
        # compare it at iteration 0 only, where it comes from objective_new
        if k == 0:
            assert abs(rec["obj"] / objs[k] - 1) < 1e-3
        assert abs(rec["test_err"] - te[k][0]) < 1e-3 and abs(rec["test_ndcg"] - te[k][1]) < 1e-3
        assert abs(rec["train_err"] - tr[k][0]) < 1e-3 and abs(rec["train_ndcg"] - tr[k][1]) < 1e-3


def test_config0_toy_shaped_real_valued_k10(oracle):
    """BASELINE configs[0]: toy-example-shaped data (6040 x 3952, ~900k REAL-VALUED ratings, PrimalCR++ -k 10):
    the reference's own CPU-runnable case.  fp64 on the GPU vs the oracle: one outer iteration function by
    function (real-valued ratings are bucketed by lround into ~9 levels; k = 10 takes the 4-lane row path)."""
    import primalcr_amd as pcr
    from primalcr_amd import synth
    R = synth.generate("toy")
    r, lam = 10, 5000.0
    X = oracle.build_csr(R.d1, R.d2, R.user, R.item, R.val)
    U0, V0 = pcr.initial(R.d1, r), pcr.initial(R.d2, r)
    s = pcr.Solver(pcr.Dataset.from_ratings(R), pcr.Parameter(k=r, precision=pcr.PCR_F64, **{"lambda": lam}))
    s.set_factors(U0, V0)
    m = s.comp_m()
    mo = oracle.comp_m(U0, V0, X)
    assert np.abs(m - mo).max() < 1e-12
    assert abs(s.objective() / oracle.objective_new(mo, U0, V0, X, lam) - 1) < 1e-11
    g, go = s.obtain_g(), oracle.obtain_g_new(U0, V0, X, mo, lam)
    assert np.abs(g - go).max() / np.abs(go).max() < 1e-10
    V1, m1, objV, info_o = oracle.update_V_new(X, lam, 1.0, U0, V0)
    oV, info = s.update_V()
    assert info["cg"] == info_o["cg"] and info["ls"] == info_o["ls"]
    assert abs(oV / objV - 1) < 1e-9
    U1, objU, iu_o = oracle.update_U_new(X, m1, lam, 1.0, V1, U0)
    oU, iu = s.update_U()
    Ug, Vg = s.get_factors()
    assert np.abs(Vg - V1).max() / np.abs(V1).max() < 1e-7 and np.abs(Ug - U1).max() / np.abs(U1).max() < 1e-7
    assert abs(oU / objU - 1) < 1e-9 and iu["cg"] == iu_o["cg"] and iu["ls"] == iu_o["ls"]


def test_config2_solver1_ml1m_vs_reference_binary(tmp_path):
    """BASELINE configs[2]: ml1m-shaped, PrimalCR (solver 1) -k 100: default precision on the GPU vs the unmodified
    reference binary (-s 1), 2 outer iterations: NDCG@10 / pairwise error within 1e-3."""
    import os
    import re
    import subprocess
    import primalcr_amd as pcr
    from oracle import oracle_py
    from primalcr_amd import synth
    if not os.path.exists(oracle_py.REF_TRAIN):
        pytest.skip("oracle/_ref was not built (no /root/reference at build time)")
    R = synth.generate("ml1m")
    d = synth.write_dir(R, str(tmp_path / "ml1m"))
    iters = 2
    out = subprocess.run([oracle_py.REF_TRAIN, "-s", "1", "-k", "100", "-l", "5000", "-t", str(iters), "-n", "16",
                          d, str(tmp_path / "ref.model")], cwd=tmp_path, capture_output=True, text=True, check=True).stdout
    te = [(float(a), float(b)) for a, b in re.findall(r"^\(Testing\) pairwise error is (\S+) and ndcg is (\S+)$", out, re.M)]
    tr = [(float(a), float(b)) for a, b in re.findall(r"^\(Training\) pairwise error is (\S+) and ndcg is (\S+)$", out, re.M)]
    s = pcr.Solver(pcr.Dataset.load(d), pcr.Parameter(k=100, solver_type=1, maxiter=iters, **{"lambda": 5000.0}))
    s.set_factors(pcr.initial(R.d1, 100), pcr.initial(R.d2, 100))
    recs, lines = s.train()
    assert lines[0] == "running PrimalCR ndcg_k is 10"
    for k, rec in enumerate(recs):
        assert abs(rec["test_err"] - te[k][0]) < 1e-3 and abs(rec["test_ndcg"] - te[k][1]) < 1e-3
        assert abs(rec["train_err"] - tr[k][0]) < 1e-3 and abs(rec["train_ndcg"] - tr[k][1]) < 1e-3


def test_half_steps_match_oracle_on_a_long_tailed_shape(oracle):
    """A Netflix-shaped slice (8000 users x 17770 items, 1.6 M ratings, users up to > 4096 ratings): every length class
    in its throughput form, the cluster class, global-scratch users, 8 SpMM tiles and the state hand-over from the U step
    to the next V step.  fp64: every half step must agree with the oracle to summation-order rounding, with identical
    CG / line-search counts."""
    import primalcr_amd as pcr
    from primalcr_amd import synth
    R = synth.generate("netflix", d1=8000, nnz=1600000)
    X = oracle.build_csr(R.d1, R.d2, R.user, R.item, R.val)
    k, lam = 16, 5000.0
    U = oracle.initial(R.d1, k); V = oracle.initial(R.d2, k)
    s = pcr.Solver(pcr.Dataset.from_ratings(R), pcr.Parameter(k=k, precision=pcr.PCR_F64, **{"lambda": lam}))
    s.set_factors(U, V)
    rel = lambda x, y: float(np.abs(x - y).max() / np.abs(y).max())
    for _ in range(2):
        V, m, objV, iv = oracle.update_V_new(X, lam, 1.0, U, V)
        gV, giv = s.update_V()
        assert abs(gV / objV - 1) < 1e-11 and (giv["cg"], giv["ls"]) == (iv["cg"], iv["ls"])
        assert rel(s.get_factors()[1], V) < 1e-10
        U, objU, iu = oracle.update_U_new(X, m, lam, 1.0, V, U)
        gU, giu = s.update_U()
        assert abs(gU / objU - 1) < 1e-11 and (giu["cg"], giu["ls"]) == (iu["cg"], iu["ls"])
        assert rel(s.get_factors()[0], U) < 1e-10


def test_setup_phases_of_solver_creation_are_kept():
    """pcr_solver_setup_phase (the split `omp-pmf-train --timing` prints as [timing-create]): the phases of pcr_solver_create in
    order, each with its wall time; their sum is the creation's own time, the list ends with PCR_ERR_ARG and no error text."""
    import time
    import primalcr_amd as pcr
    from primalcr_amd import synth
    R = synth.generate("small", seed=4)
    ds = pcr.Dataset.from_ratings(R)
    t0 = time.perf_counter()
    s = pcr.Solver(ds, pcr.Parameter(k=8, **{"lambda": 10.0}))
    wall_ms = 1e3 * (time.perf_counter() - t0)
    ph = s.setup_phases()
    names = [n for n, _ in ph]
    assert names[0] == "device, streams, events" and names[-1] == "stream lanes" and "tile-major CSC, slab plan" in names and len(names) >= 8
    assert all(ms >= 0.0 for _, ms in ph) and 0.5 * wall_ms < sum(ms for _, ms in ph) <= wall_ms + 1.0
    s.close()

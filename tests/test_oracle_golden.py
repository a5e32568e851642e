"""CPU tests: pin the C restatement (oracle/pcr_oracle.c) against golden vectors dumped from
the unmodified reference (oracle/make_golden.py -> tests/golden/).  No GPU, no /root/reference."""
import os
import re

import numpy as np
import pytest

from conftest import GOLDEN_CASES, ROOT, golden_csr, load_golden

RT = 1e-11   # the restatement follows the reference's loop order; only sort-tie order may differ


def rel(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)) if a.size else 0.0


def test_initial_known_draws(oracle):
    # first draws of libstdc++ minstd_rand0 + normal_distribution<double>
    X = oracle.initial(2, 2).ravel()
    assert X[0] == -0.12196578414159691
    assert X[1] == -1.0868180442613573
    assert X[2] == 0.68428994379655483


@pytest.mark.parametrize("name", GOLDEN_CASES)
def test_initial_and_csr(oracle, name):
    g, _ = load_golden(name)
    d1, d2, r = int(g["d1"]), int(g["d2"]), int(g["r"])
    U = oracle.initial(d1, r)
    assert np.array_equal(U, g["U0"])
    V = oracle.initial(d2, r)
    if name != "edge5":                       # edge5 overwrites V0[6] with V0[5]
        assert np.array_equal(V, g["V0"])
    # quirk q1: U and V share the stream
    n = min(d1, d2)
    assert np.array_equal(U[:n], V[:n])
    X = oracle.build_csr(d1, d2, g["user"], g["item"], g["val"])
    assert np.array_equal(X.idx, g["csr_idx"]) and np.array_equal(X.item, g["csr_item"])
    assert np.array_equal(X.val, g["csr_val"])
    XT = oracle.build_csr_test(d1, d2, g["tuser"], g["titem"], g["tval"])
    assert np.array_equal(XT.idx, g["tcsr_idx"]) and np.array_equal(XT.item, g["tcsr_item"])


@pytest.mark.parametrize("name", GOLDEN_CASES)
def test_v_side_functions(oracle, name):
    g, _ = load_golden(name)
    X = golden_csr(g); U, V, lam = g["U0"], g["V0"], float(g["lam"])
    m = oracle.comp_m(U, V, X)
    assert rel(m, g["m"]) <= 1e-15
    assert abs(oracle.objective_new(m, U, V, X, lam) / float(g["obj"]) - 1) < RT
    assert rel(oracle.obtain_g_new(U, V, X, m, lam), g["g"]) < RT
    assert rel(oracle.compute_Ha_new(g["a"], m, U, X, lam), g["Ha"]) < RT
    delta, _ = oracle.solve_delta_new(g["g"], m, U, X, lam)
    assert rel(delta, g["delta"]) < 1e-9
    V1, m1, objV, info = oracle.update_V_new(X, lam, 1.0, U, V)
    assert rel(V1, g["V1"]) < 1e-9 and rel(m1, g["m1"]) < 1e-9
    assert abs(objV / float(g["objV"]) - 1) < 1e-10
    assert info["accepted"] == 1


@pytest.mark.parametrize("name", GOLDEN_CASES)
def test_u_side_functions(oracle, name):
    g, _ = load_golden(name)
    X = golden_csr(g); U, lam = g["U0"], float(g["lam"])
    U1, objU, _ = oracle.update_U_new(X, g["m1"], lam, 1.0, g["V1"], U)
    assert rel(U1, g["U1"]) < 1e-9
    assert abs(objU / float(g["objU"]) - 1) < 1e-10
    for k, i in enumerate(g["u_users"]):
        un, ob, _ = oracle.update_u_new(int(i), g["V1"], X, g["m1"], lam, 1.0, U[i])
        assert np.abs(un - g["u_new"][k]).max() < 1e-9 * max(1.0, np.abs(g["u_new"][k]).max())
        assert abs(ob - g["u_obj"][k]) <= 1e-10 * max(1.0, abs(g["u_obj"][k]))


@pytest.mark.parametrize("name", GOLDEN_CASES)
def test_solver1_functions(oracle, name):
    g, _ = load_golden(name)
    X = golden_csr(g); U, V, lam = g["U0"], g["V0"], float(g["lam"])
    m = g["m"]
    assert abs(oracle.objective_new(m, U, V, X, lam, solver=1) / float(g["obj_s1"]) - 1) < RT
    assert rel(oracle.obtain_g_new(U, V, X, m, lam, solver=1), g["g_s1"]) < RT
    assert rel(oracle.compute_Ha_new(g["a"], m, U, X, lam, solver=1), g["Ha_s1"]) < RT
    for k, i in enumerate(g["u_users"]):
        un, ob, _ = oracle.update_u_new(int(i), g["V1"], X, g["m1"], lam, 1.0, U[i], solver=1)
        assert np.abs(un - g["u_new_s1"][k]).max() < 1e-9 * max(1.0, np.abs(g["u_new_s1"][k]).max())
        assert abs(ob - g["u_obj_s1"][k]) <= 1e-10 * max(1.0, abs(g["u_obj_s1"][k]))
    if name != "real":
        # sweep == brute force on integer ratings
        assert abs(float(g["obj"]) / float(g["obj_s1"]) - 1) < 1e-12
        assert rel(g["g"], g["g_s1"]) < 1e-11 and rel(g["Ha"], g["Ha_s1"]) < 1e-11


@pytest.mark.parametrize("name", GOLDEN_CASES)
def test_evaluator(oracle, name):
    g, _ = load_golden(name)
    X, XT = golden_csr(g), golden_csr(g, test=True)
    for tag, U, V in (("eval0", g["U0"], g["V0"]), ("eval1", g["U1"], g["V1"])):
        if name == "edge5" and tag == "eval0":
            continue                         # V0[6]==V0[5]: exact score ties, NDCG tie order unspecified
        assert np.allclose(oracle.eval(U, V, X), g[tag + "_train"], rtol=0, atol=1e-12)
        assert np.allclose(oracle.eval(U, V, XT), g[tag + "_test"], rtol=0, atol=1e-12)
    # ties count as errors regardless of order: pairwise error is pinned even with duplicate scores
    assert abs(oracle.eval(g["U0"], g["V0"], X)[0] - g["eval0_train"][0]) < 1e-12


def parse_stdout(text):
    objs = [float(x) for x in re.findall(r"^Iter \d+ time \S+ obj (\S+)$", text, re.M)]
    tr = [(float(a), float(b)) for a, b in re.findall(r"^\(Training\) pairwise error is (\S+) and ndcg is (\S+)$", text, re.M)]
    te = [(float(a), float(b)) for a, b in re.findall(r"^\(Testing\) pairwise error is (\S+) and ndcg is (\S+)$", text, re.M)]
    return objs, tr, te


@pytest.mark.parametrize("name", GOLDEN_CASES)
@pytest.mark.parametrize("solver", [2, 1])
def test_end_to_end_vs_reference_cli(oracle, name, solver):
    """orc_train from the reference's init must reproduce omp-pmf-train -n 1: printed objective /
    pairwise error / NDCG (6 significant digits) and the model file's U, V."""
    g, meta = load_golden(name)
    d1, d2, r, lam = int(g["d1"]), int(g["d2"]), int(g["r"]), float(g["lam"])
    X, XT = golden_csr(g), golden_csr(g, test=True)
    U0, V0 = oracle.initial(d1, r), oracle.initial(d2, r)
    U, V, recs = oracle.train(X, U0, V0, lam, meta["iters"], XT, solver=solver)
    objs, tr, te = parse_stdout(meta[f"stdout_s{solver}"])
    assert len(objs) == meta["iters"] + 1
    for k, rec in enumerate(recs):
        assert abs(rec["obj"] / objs[k] - 1) < 2e-5            # printed with 6 significant digits
        assert abs(rec["train_err"] - tr[k][0]) < 2e-6 and abs(rec["train_ndcg"] - tr[k][1]) < 2e-6
        assert abs(rec["test_err"] - te[k][0]) < 2e-6 and abs(rec["test_ndcg"] - te[k][1]) < 2e-6
    assert rel(U, g[f"cli_U_s{solver}"]) < 1e-8 and rel(V, g[f"cli_V_s{solver}"]) < 1e-8
    assert meta[f"model_bytes_s{solver}"] == 2 * 16 + 8 * r * (d1 + d2)


def test_known_answer_zero_model_counts_pairs(oracle):
    # objective at U = V = 0 equals #Omega
    g, _ = load_golden("mid5")
    X = golden_csr(g)
    U = np.zeros_like(g["U0"]); V = np.zeros_like(g["V0"])
    m = oracle.comp_m(U, V, X)
    assert oracle.objective_new(m, U, V, X, 5000.0) == float(g["n_pairs"]) == oracle.count_pairs(X)
    assert oracle.objective_new(m, U, V, X, 5000.0, solver=1) == float(g["n_pairs"])


def _shipped_case(name="ml1m_test"):
    import json
    g = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
    meta = json.load(open(os.path.join(ROOT, "tests", "golden", name + ".json")))
    return g["user"].astype(np.int64), g["item"].astype(np.int64), g["val"].astype(np.float64), meta


def _printed(stdout):
    import re
    it = [float(x) for x in re.findall(r"^Iter \d+ time \S+ obj (\S+)$", stdout, re.M)]
    ev = [(t, float(a), float(b)) for t, a, b in re.findall(r"^\((Training|Testing)\) pairwise error is (\S+) and ndcg is (\S+)$", stdout, re.M)]
    return it, ev


@pytest.mark.parametrize("name,tag,solver,lam", [("ml1m_test", "s2_l5000", 2, 5000.0), ("ml1m_test", "s2_l50", 2, 50.0),
                                                 ("ml1m_test", "s1_l50", 1, 50.0), ("toy_test", "s2", 2, 5000.0), ("toy_test", "s1", 1, 5000.0)])
def test_oracle_on_the_references_own_rating_files(oracle, name, tag, solver, lam):
    """The rating files the reference ships, used as training and test set, as printed by the unmodified binary with -n 1:
    ml1m/test.ratings (real MovieLens ratings; the known answers of BASELINE.md section 2: objective 187 644 = #Omega at
    lambda 5000, NDCG@10 0.979346 at lambda 50) and toy-example/test.ratings (real-valued ratings: 9 lround levels,
    non-positive gains, NDCG beyond 1 -- configs[0])."""
    user, item, val, meta = _shipped_case(name)
    X = oracle.build_csr(meta["d1"], meta["d2"], user, item, val)
    U0 = oracle.initial(meta["d1"], meta["k"]); V0 = oracle.initial(meta["d2"], meta["k"])
    _, _, recs = oracle.train(X, U0, V0, lam, meta["iters"], XT=X, solver=solver, do_predict=1)
    objs, evs = _printed(meta["stdout"][tag])
    assert len(objs) == meta["iters"] + 1
    for r, o in zip(recs, objs):
        assert abs(r["obj"] / o - 1) < 6e-6                       # 6 printed digits (%g)
    for r, (tr, te) in zip(recs, zip(evs[0::2], evs[1::2])):
        assert abs(r["train_err"] - tr[1]) < 6e-6 and abs(r["train_ndcg"] - tr[2]) < 6e-6
        assert abs(r["test_err"] - te[1]) < 6e-6 and abs(r["test_ndcg"] - te[2]) < 6e-6

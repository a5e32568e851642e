"""The drop-in CLIs primalcr_amd/bin/omp-pmf-train and omp-pmf-predict (pmf-train.cpp, pmf-predict.cpp).

CPU part: usage text, exit codes, option handling and I/O failures (nothing reaches the GPU).
GPU part (-m gpu): end-to-end runs against the golden stdout / model / predictions of the unmodified
reference on the same data directories.
"""
import os
import re
import struct
import subprocess

import numpy as np
import pytest

from conftest import BIN_DIR, GOLDEN_CASES, ROOT, load_golden
from primalcr_amd import synth

TRAIN = os.path.join(BIN_DIR, "omp-pmf-train")
PREDICT = os.path.join(BIN_DIR, "omp-pmf-predict")
NUM = r"[-+]?(?:\d+\.?\d*|\.\d+)(?:e[-+]?\d+)?"


def run(cmd, cwd):
    return subprocess.run(cmd, cwd=cwd, capture_output=True, text=True)


def test_usage_and_exit_codes(tmp_path):
    r = run([TRAIN], tmp_path)                                    # pmf-train.cpp:115-116
    assert r.returncode == 1 and r.stdout.startswith("Usage: omp-pmf-train [options] data_dir [model_filename]")
    for flag in ("-s type", "-k rank", "-n threads", "-l lambda", "-t max_iter", "-p do_predict", "--cache file", "--snapshot-every n"):
        assert flag in r.stdout
    r = run([TRAIN, "-z", "3", "dir"], tmp_path)                  # pmf-train.cpp:104-107
    assert r.returncode == 1 and "unknown option: -z" in r.stderr
    r = run([TRAIN, "-k"], tmp_path)                              # option without value
    assert r.returncode == 1
    r = run([TRAIN, "-s", "7", "dir"], tmp_path)                  # pmf-train.cpp:331-333: message, exit 0
    assert r.returncode == 0 and "wrong solver type (7)" in r.stderr
    r = run([TRAIN, str(tmp_path / "missing_dir"), "m.model"], tmp_path)
    assert r.returncode == 1 and "can't open" in r.stderr          # the reference segfaults here (util.cpp:10)
    r = run([TRAIN, "data", str(tmp_path / "no" / "such" / "m.model")], tmp_path)
    assert r.returncode == 1 and "can't open output file" in r.stderr   # pmf-train.cpp:254-258
    r = run([PREDICT], tmp_path)
    assert r.returncode == 1 and r.stdout.startswith("Usage: omp-pmf-predict test_file model output_file")
    r = run([PREDICT, "nope", "m", "o"], tmp_path)
    assert r.returncode == 1 and "can't open test file nope" in r.stderr


def test_default_model_name_rule(tmp_path):
    """pmf-train.cpp:120-133: model = basename(data_dir) + '.model' in cwd, trailing slashes stripped;
    the model file is opened before the data is read (pmf-train.cpp:252-259)."""
    r = run([TRAIN, "some/dir/mydata///"], tmp_path)
    assert r.returncode == 1                                       # data dir does not exist ...
    assert (tmp_path / "mydata.model").exists()                    # ... but the model file was created first


def test_predict_on_the_host_needs_no_gpu(tmp_path):
    """BASELINE configs[0] ("plumbing, runs without a GPU"), predict half: `omp-pmf-predict --host` scores with the reference's
    own loop (one fp64 dot product per line, pmf-predict.cpp:56-64) -- the same bytes as the reference binary writes -- and
    only when asked: without --host a machine without a GPU gets an error, never a silent CPU path."""
    import primalcr_amd as pcr
    from oracle import oracle_py
    R = synth.generate("tiny")
    d = synth.write_dir(R, str(tmp_path / "data"))
    rng = np.random.default_rng(3)
    U, V = rng.normal(size=(R.d1, 7)), rng.normal(size=(R.d2, 7))
    pcr.model_save(str(tmp_path / "m.model"), U, V)
    test_file = os.path.join(d, "test.ratings")
    r = run([PREDICT, "--host", test_file, "m.model", "ours.txt"], tmp_path)
    assert r.returncode == 0, r.stderr
    ours = open(tmp_path / "ours.txt").read()
    want = "".join("%f\n" % float(U[u] @ V[i]) for u, i in zip(R.tuser, R.titem))
    got = [float(x) for x in ours.split()]
    assert len(got) == len(R.tuser) and np.allclose(got, [float(U[u] @ V[i]) for u, i in zip(R.tuser, R.titem)], atol=1e-6)
    if os.path.exists(oracle_py.REF_PREDICT):
        ref = run([oracle_py.REF_PREDICT, test_file, "m.model", "ref.txt"], tmp_path)
        assert ref.returncode == 0 and open(tmp_path / "ref.txt").read() == ours          # byte for byte
    else:
        assert ours == want
    bad = run([PREDICT, "--host", test_file, "m.model"], tmp_path)                         # usage is unchanged
    assert bad.returncode == 1 and bad.stdout.startswith("Usage: omp-pmf-predict test_file model output_file")
    import torch
    if not torch.cuda.is_available():
        r = run([PREDICT, test_file, "m.model", "gpu.txt"], tmp_path)
        assert r.returncode == 1 and "no HIP device" in r.stderr


def golden_dir(name, tmp_path):
    g, meta = load_golden(name)
    R = synth.Ratings(int(g["d1"]), int(g["d2"]), g["user"], g["item"], g["val"], g["tuser"], g["titem"], g["tval"])
    return g, meta, synth.write_dir(R, str(tmp_path / "data"))


@pytest.mark.gpu
@pytest.mark.parametrize("solver", [2, 1])
@pytest.mark.parametrize("name", GOLDEN_CASES)
def test_train_cli_matches_reference_end_to_end(name, solver, tmp_path):
    g, meta, d = golden_dir(name, tmp_path)
    lam, r = float(g["lam"]), int(g["r"])
    out = run([TRAIN, "-s", str(solver), "-k", str(r), "-n", "1", "-l", repr(lam), "-t", str(meta["iters"]), "--f64",
               d, "m.model"], tmp_path)
    assert out.returncode == 0, out.stderr
    ours = [re.sub(r"time \S+", "time T", l) for l in out.stdout.strip().split("\n") if not l.startswith("Wall-time")]
    theirs = [re.sub(r"time \S+", "time T", l) for l in meta[f"stdout_s{solver}"].strip().split("\n") if not l.startswith("Wall-time")]
    assert len(ours) == len(theirs)
    for a, b in zip(ours, theirs):
        if a != b:
            assert re.sub(NUM, "#", a) == re.sub(NUM, "#", b), (a, b)
            assert np.allclose([float(x) for x in re.findall(NUM, a)], [float(x) for x in re.findall(NUM, b)], rtol=2e-5, atol=2e-6)
    raw = open(tmp_path / "m.model", "rb").read()
    assert len(raw) == meta[f"model_bytes_s{solver}"]
    d1, k = struct.unpack("ll", raw[:16])
    U = np.frombuffer(raw, np.float64, d1 * k, 16).reshape(d1, k)
    off = 16 + 8 * d1 * k
    d2, k2 = struct.unpack("ll", raw[off:off + 16])
    V = np.frombuffer(raw, np.float64, d2 * k2, off + 16).reshape(d2, k2)
    assert np.abs(U - g[f"cli_U_s{solver}"]).max() < 1e-6 * np.abs(U).max()
    assert np.abs(V - g[f"cli_V_s{solver}"]).max() < 1e-6 * np.abs(V).max()
    side = "U.txt" if solver == 2 else f"U{int(lam)}.txt"            # pmf-train.cpp:208, 277
    assert (tmp_path / side).exists() and (tmp_path / side.replace("U", "V")).exists()
    assert len(open(tmp_path / side).read().strip().split("\n")) == d1
    # the reference writes `f << U[i][j]` with an ofstream at its default precision (pmf-train.cpp:276-295) = C's "%g"
    assert open(tmp_path / side).read() == "".join(" ".join("%g" % x for x in row) + "\n" for row in U)
    assert open(tmp_path / side.replace("U", "V")).read() == "".join(" ".join("%g" % x for x in row) + "\n" for row in V)
    # predict: one "%lf" per test line (pmf-predict.cpp:63)
    p = run([PREDICT, os.path.join(d, "test.ratings"), "m.model", "pred.txt"], tmp_path)
    assert p.returncode == 0, p.stderr
    ours_p = np.array([float(x) for x in open(tmp_path / "pred.txt").read().split()])
    ref_p = np.array([float(x) for x in meta[f"predict_s{solver}"].split()])
    assert ours_p.shape == ref_p.shape and np.abs(ours_p - ref_p).max() < 2e-6


@pytest.mark.gpu
def test_timing_flag_splits_the_wall_time_and_large_side_files_keep_the_format(tmp_path):
    """--timing (round-4 verdict item 2: where does the drop-in CLI's wall time go?): one "[timing] ..." line on stderr whose
    phases are non-negative, add up to no more than the process's wall time, and whose iter_s is the reference's own "Iter k
    time" clock.  On an ml1m-sized model the side files are formatted by several threads: still the reference's bytes."""
    R = synth.generate("ml1m", d1=3000, nnz=400_000)
    d = synth.write_dir(R, str(tmp_path / "data"))
    out = run([TRAIN, "-k", "100", "-t", "2", "--timing", d, "m.model"], tmp_path)
    assert out.returncode == 0, out.stderr
    m = re.search(r"^\[timing\] (.*)$", out.stderr, re.M)
    assert m, out.stderr
    ph = {k: float(v) for k, v in (kv.split("=") for kv in m.group(1).split())}
    assert set(ph) == {"load_s", "init_s", "create_s", "train_s", "iter_s", "eval_s", "write_s", "wall_s"} and min(ph.values()) >= 0
    assert ph["load_s"] + ph["init_s"] + ph["create_s"] + ph["train_s"] + ph["write_s"] <= ph["wall_s"] + 1e-3
    assert ph["eval_s"] == pytest.approx(ph["train_s"] - ph["iter_s"], abs=2e-4) and ph["eval_s"] > 0          # -p 1 is the default
    last = float(re.findall(r"^Iter 2 time (\S+) obj", out.stdout, re.M)[-1])
    assert ph["iter_s"] == pytest.approx(last, rel=1e-3, abs=1e-4)
    raw = open(tmp_path / "m.model", "rb").read()
    d1, k = struct.unpack("ll", raw[:16])
    U = np.frombuffer(raw, np.float64, d1 * k, 16).reshape(d1, k)
    assert d1 * k > 4 * 65536                                                                                # several formatting threads
    assert open(tmp_path / "U.txt").read() == "".join(" ".join("%g" % x for x in row) + "\n" for row in U)
    quiet = run([TRAIN, "-k", "10", "-t", "1", "-p", "0", d, "q.model"], tmp_path)
    assert quiet.returncode == 0 and "[timing]" not in quiet.stderr


@pytest.mark.gpu
def test_train_cli_default_precision_quality(tmp_path):
    """Default (fp32 storage) CLI run: NDCG / pairwise error lines within 1e-3 of the reference's."""
    g, meta, d = golden_dir("mid5", tmp_path)
    out = run([TRAIN, "-k", str(int(g["r"])), "-l", repr(float(g["lam"])), "-t", str(meta["iters"]), d], tmp_path)
    assert out.returncode == 0, out.stderr
    assert (tmp_path / "data.model").exists()
    pat = r"^\((Training|Testing)\) pairwise error is (\S+) and ndcg is (\S+)$"
    a = re.findall(pat, out.stdout, re.M); b = re.findall(pat, meta["stdout_s2"], re.M)
    assert len(a) == len(b) > 0
    for (t1, e1, n1), (t2, e2, n2) in zip(a, b):
        assert t1 == t2 and abs(float(e1) - float(e2)) < 1e-3 and abs(float(n1) - float(n2)) < 1e-3


@pytest.mark.gpu
def test_warm_start_continues_the_trajectory(tmp_path):
    """--init-model: 2 iterations, then 1 more from the saved model == 3 iterations in one run
    (fp64: every outer iteration recomputes m and resets the step size, so the state is just U and V)."""
    g, meta, d = golden_dir("mid5", tmp_path)
    base = [TRAIN, "-k", str(int(g["r"])), "-l", repr(float(g["lam"])), "-p", "0", "--f64"]
    assert run(base + ["-t", "3", d, "m3.model"], tmp_path).returncode == 0
    assert run(base + ["-t", "2", d, "m2.model"], tmp_path).returncode == 0
    r = run(base + ["-t", "1", "--init-model", "m2.model", d, "m21.model"], tmp_path)
    assert r.returncode == 0, r.stderr
    a = np.frombuffer(open(tmp_path / "m3.model", "rb").read(), np.float64)
    b = np.frombuffer(open(tmp_path / "m21.model", "rb").read(), np.float64)
    assert a.shape == b.shape and np.nanmax(np.abs(a[2:] - b[2:])) < 1e-9 * np.nanmax(np.abs(a[2:]))
    bad = run(base + ["-k", "3", "-t", "1", "--init-model", "m2.model", d, "x.model"], tmp_path)
    assert bad.returncode == 1 and "expected" in bad.stderr


@pytest.mark.gpu
def test_cache_and_snapshots(tmp_path):
    """--cache gives the same run as the text parse (first call writes it, second reads it);
    --snapshot-every 2 writes <model>.iter2 = the model of a -t 2 run, .iter4 = the final model."""
    g, meta, d = golden_dir("mid5", tmp_path)
    base = [TRAIN, "-k", str(int(g["r"])), "-l", repr(float(g["lam"])), "-p", "0", "--f64"]
    cache = str(tmp_path / "mid5.cache")
    plain = run(base + ["-t", "2", d, "plain.model"], tmp_path)
    first = run(base + ["-t", "2", "--cache", cache, d, "c1.model"], tmp_path)
    assert plain.returncode == 0 and first.returncode == 0 and os.path.exists(cache)
    second = run(base + ["-t", "4", "--cache", cache, "--snapshot-every", "2", d, "c2.model"], tmp_path)
    assert second.returncode == 0, second.stderr
    rd = lambda n: open(tmp_path / n, "rb").read()
    assert rd("plain.model") == rd("c1.model") == rd("c2.model.iter2")
    assert rd("c2.model.iter4") == rd("c2.model")
    assert not (tmp_path / "c2.model.iter1").exists() and not (tmp_path / "c2.model.iter3").exists()
    iters = lambda s: [re.sub(r"time \S+", "time T", l) for l in s.split("\n") if l.startswith("Iter ")]
    assert iters(plain.stdout) == iters(first.stdout) == iters(second.stdout)[:3]


@pytest.mark.gpu
@pytest.mark.parametrize("name,tag,args", [("ml1m_test", "s2_l5000", ["-s", "2", "-l", "5000"]), ("ml1m_test", "s2_l50", ["-s", "2", "-l", "50"]),
                                           ("ml1m_test", "s1_l50", ["-s", "1", "-l", "50"]), ("toy_test", "s2", ["-s", "2"]), ("toy_test", "s1", ["-s", "1"])])
def test_train_cli_on_the_references_own_rating_files(name, tag, args, tmp_path):
    """The rating files the reference ships (ml1m/test.ratings: real MovieLens ratings; toy-example/test.ratings: real-valued
    ratings, configs[0]) as training and test set: our omp-pmf-train --f64 must print the lines the unmodified reference prints
    (objective 187 644 = #Omega at lambda 5000, NDCG@10 0.979346 at lambda 50) to the 6 printed digits."""
    import json
    g = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
    meta = json.load(open(os.path.join(ROOT, "tests", "golden", name + ".json")))
    u, i, v = g["user"].astype(np.int32), g["item"].astype(np.int32), g["val"].astype(np.float64)
    R = synth.Ratings(meta["d1"], meta["d2"], u, i, v, u, i, v)
    d = synth.write_dir(R, str(tmp_path / "data"))
    out = run([TRAIN, *args, "-k", str(meta["k"]), "-n", "1", "-t", str(meta["iters"]), "-p", "1", "--f64", d, "m.model"], tmp_path)
    assert out.returncode == 0, out.stderr
    pick = lambda s: [l for l in s.split("\n") if l.startswith(("Iter", "(T"))]
    ours, theirs = pick(out.stdout), pick(meta["stdout"][tag])
    assert len(ours) == len(theirs) == 3 * (meta["iters"] + 1)
    # toy_test: 112 of the 1500 users have all their (real-valued) ratings in ONE lround bucket: no comparable pair, PrimalCR++
    # drives their u to ~1e-17 of rounding noise, and the evaluator -- which compares RAW ratings -- then ranks their 2-4 items
    # by the sign of that noise.  The reference's printed metrics contain that noise (the oracle reproduces it only because it
    # repeats the reference's operation order bit for bit); any other summation order moves them by up to ~112/1500 * 1/2.
    # PrimalCR keeps those users' u (cc == 0, pcr.cpp:552), so its lines and every objective line are compared exactly.
    noisy = name == "toy_test" and tag == "s2"
    for a, b in zip(ours, theirs):
        a = re.sub(r"time \S+", "time T", a); b = re.sub(r"time \S+", "time T", b)
        assert re.sub(NUM, "#", a) == re.sub(NUM, "#", b), (a, b)
        tol = 4e-2 if (noisy and a.startswith("(T")) else 6e-6
        assert np.allclose([float(x) for x in re.findall(NUM, a)], [float(x) for x in re.findall(NUM, b)], rtol=tol, atol=tol), (a, b)
    if noisy:
        # ... so this configuration's metrics are pinned here WITHOUT those users: the model our CLI wrote and the model of the
        # oracle's own training run (bit-for-bit the reference's: tests/test_oracle_golden.py), both evaluated by the oracle's
        # evaluator over the users that do have a comparable pair, must agree to 1e-6 -- and the noise-ranked users' rows must
        # have been driven to (rounding) zero in both.
        from oracle.oracle_py import Oracle
        orc = Oracle()
        raw = open(tmp_path / "m.model", "rb").read()
        d1, k = struct.unpack("ll", raw[:16])
        U = np.frombuffer(raw, np.float64, d1 * k, 16).reshape(d1, k)
        off = 16 + 8 * d1 * k
        d2, _ = struct.unpack("ll", raw[off:off + 16])
        V = np.frombuffer(raw, np.float64, d2 * k, off + 16).reshape(d2, k)
        X = orc.build_csr(meta["d1"], meta["d2"], u, i, v)
        Uo, Vo, _ = orc.train(X, orc.initial(meta["d1"], k), orc.initial(meta["d2"], k), meta["lam"], meta["iters"], solver=2, do_predict=0)
        lv = np.rint(v).astype(np.int64)
        lo = np.full(meta["d1"], np.iinfo(np.int64).max); hi = np.full(meta["d1"], np.iinfo(np.int64).min)
        np.minimum.at(lo, u, lv); np.maximum.at(hi, u, lv)
        flat = np.flatnonzero((hi == lo) & (np.bincount(u, minlength=meta["d1"]) > 0))   # one lround bucket: no comparable pair
        assert len(flat) == 112
        assert np.abs(U[flat]).max() < 1e-12 and np.abs(Uo[flat]).max() < 1e-12
        keep = ~np.isin(u, flat)
        XT = orc.build_csr_test(meta["d1"], meta["d2"], u[keep], i[keep], v[keep])
        e_ours, n_ours = orc.eval(U, V, XT)
        e_ref, n_ref = orc.eval(Uo, Vo, XT)
        assert abs(e_ours - e_ref) < 1e-6 and abs(n_ours - n_ref) < 1e-6 * max(1.0, abs(n_ref)), (e_ours, e_ref, n_ours, n_ref)


def test_predict_reads_and_writes_like_the_reference_at_size(tmp_path):
    """omp-pmf-predict parses its test file with the host threads (several pieces above 1 MB) and formats its output in threads:
    150 000 lines through `--host` must be the reference binary's bytes, and like the reference (one fscanf per line until it
    fails) a malformed line ends the input: the lines before it are scored, nothing after."""
    import primalcr_amd as pcr
    from oracle import oracle_py
    rng = np.random.default_rng(8)
    d1, d2, k, n = 5000, 3000, 9, 150_000
    U, V = rng.normal(size=(d1, k)) * 3, rng.normal(size=(d2, k)) * 3
    U[7] *= 1e5; V[11] *= 1e-7                                       # large and tiny scores: the fixed-notation formatter's corners
    pcr.model_save(str(tmp_path / "m.model"), U, V)
    user, item = rng.integers(0, d1, n), rng.integers(0, d2, n)
    lines = [f"{u + 1} {i + 1} {1 + (u + i) % 5}" for u, i in zip(user.tolist(), item.tolist())]
    (tmp_path / "t.ratings").write_text("\n".join(lines) + "\n")
    assert os.path.getsize(tmp_path / "t.ratings") > (1 << 20)
    r = run([PREDICT, "--host", "t.ratings", "m.model", "ours.txt"], tmp_path)
    assert r.returncode == 0, r.stderr
    ours = open(tmp_path / "ours.txt").read()
    assert ours.count("\n") == n
    if os.path.exists(oracle_py.REF_PREDICT):
        ref = run([oracle_py.REF_PREDICT, "t.ratings", "m.model", "ref.txt"], tmp_path)
        assert ref.returncode == 0 and open(tmp_path / "ref.txt").read() == ours
    else:
        assert ours == "".join("%f\n" % float(U[u] @ V[i]) for u, i in zip(user, item))
    bad = list(lines)
    bad[100_003] = "17 oops 3"
    (tmp_path / "bad.ratings").write_text("\n".join(bad) + "\n")
    r = run([PREDICT, "--host", "bad.ratings", "m.model", "cut.txt"], tmp_path)
    assert r.returncode == 0 and open(tmp_path / "cut.txt").read() == "".join(ours.splitlines(True)[:100_003])
    # (not compared with the reference here: its loop tests `fscanf(...) != EOF`, pmf-predict.cpp:56, so a malformed token, which
    # fscanf neither consumes nor reports as EOF, makes it print the previous pair's score for ever)


def test_gpus_option_bootstrap_without_a_gpu(tmp_path):
    """--gpus N forks one worker per GPU before anything touches a GPU and the parent only waits.  On a machine without a
    GPU every worker must fail loudly in pcr_solver_create, the parent must reap them all and exit 1 -- and bad option
    values never reach the fork."""
    import torch
    R = synth.generate("tiny")
    d = synth.write_dir(R, str(tmp_path / "data"))
    r = run([TRAIN, "--gpus", "0", d, "m.model"], tmp_path)
    assert r.returncode == 1 and "--gpus must be" in r.stderr
    r = run([TRAIN, "--gpus", "2", "--devices", "0", d, "m.model"], tmp_path)
    assert r.returncode == 1 and "one ordinal per rank" in r.stderr
    r = run([TRAIN, "--gpus", "2", "--comm", "smoke-signals", d, "m.model"], tmp_path)
    assert r.returncode == 1
    r = run([TRAIN, "--tune", "nonsense=1", d, "m.model"], tmp_path)
    assert r.returncode == 1 and "unknown key" in r.stderr
    if torch.cuda.is_available():
        pytest.skip("GPU present: the failure path of the workers needs a machine without one")
    r = run([TRAIN, "--gpus", "3", "-p", "0", d, "m.model"], tmp_path)
    assert r.returncode == 1 and "a GPU worker failed" in r.stderr
    assert r.stderr.count("solver:") >= 1 and "HIP" in r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("ll", ["16", "0"], ids=["device-driven", "host-synchronised"])
@pytest.mark.parametrize("solver", [2, 1])
def test_gpus_option_two_ranks_on_one_gpu_equal_one_rank(solver, ll, tmp_path):
    """omp-pmf-train --gpus 2 --devices 0,0 --comm p2p: two worker processes share the one GPU of the test box (RCCL refuses
    two ranks on one device; the peer-to-peer communicator does not care), each owns the users pcr_partition_users gives
    its rank, every V-side vector goes through the reduce-scatter / all-gather over IPC-mapped buffers -- the device-driven
    exchange (one kernel per rank, flags inside the words; p2p_ll, the default for small vectors) or the host-synchronised
    one.  fp64: objective
    lines to the printed digits, model equal to the single-process run to summation-order rounding; --gpus 1 is the
    plain run, byte for byte."""
    import primalcr_amd as pcr
    g, meta, d = golden_dir("mid5", tmp_path)
    base = [TRAIN, "-s", str(solver), "-k", str(int(g["r"])), "-l", repr(float(g["lam"])), "-t", "3", "--f64"]
    one = run(base + [d, "one.model"], tmp_path)
    assert one.returncode == 0, one.stderr
    same = run(base + ["--gpus", "1", d, "same.model"], tmp_path)
    assert same.returncode == 0 and open(tmp_path / "same.model", "rb").read() == open(tmp_path / "one.model", "rb").read()
    two = run(base + ["--gpus", "2", "--devices", "0,0", "--comm", "p2p", "--tune", "p2p_ll=" + ll, d, "two.model"], tmp_path)
    assert two.returncode == 0, two.stderr
    strip = lambda out: [l for l in out.split("\n") if l.startswith(("Iter", "(T"))]
    la, lb = strip(one.stdout), strip(two.stdout)
    assert len(la) == len(lb) == 4 + 8
    for x, y in zip(la, lb):
        fx = [float(v) for v in re.findall(NUM, x)]; fy = [float(v) for v in re.findall(NUM, y)]
        if x.startswith("Iter"):
            fx, fy = fx[:1] + fx[2:], fy[:1] + fy[2:]              # (the time differs)
        assert re.sub(NUM, "#", x) == re.sub(NUM, "#", y) and np.allclose(fx, fy, rtol=2e-5, atol=2e-6), (x, y)
    a = np.frombuffer(open(tmp_path / "one.model", "rb").read(), np.float64)
    b = np.frombuffer(open(tmp_path / "two.model", "rb").read(), np.float64)
    assert a.shape == b.shape and np.nanmax(np.abs(a[2:] - b[2:])) < 1e-9 * np.nanmax(np.abs(a[2:]))
    # the workers' shards are the partitioner's
    idx = np.concatenate([[0], np.cumsum(np.bincount(g["user"], minlength=int(g["d1"])))])
    bounds = pcr.partition_users(idx, 2)
    shards = re.findall(r"^\[rank (\d)\] device 0: users \[(\d+), (\d+)\), (\d+) ratings$", two.stderr, re.M)
    assert sorted((int(q), int(a0), int(b0)) for q, a0, b0, _ in shards) == [(0, bounds[0], bounds[1]), (1, bounds[1], bounds[2])]


@pytest.mark.gpu
@pytest.mark.parametrize("ranks,chunks", [(2, 3), (3, 7)])
def test_gpus_option_item_ranges_overlap_the_exchange(ranks, chunks, tmp_path):
    """The SpMM item range by item range, each range's all-reduce on its own stream while the next range computes
    (pcr_tune allreduce_chunks; chosen by itself for vectors of 4 MB and more): fp64 over the peer-to-peer exchange on the one
    GPU, objective lines to the printed digits and the model to summation-order rounding against one rank."""
    g, meta, d = golden_dir("mid5", tmp_path)
    base = [TRAIN, "-k", str(int(g["r"])), "-l", repr(float(g["lam"])), "-t", "3", "--f64", "-p", "0"]
    one = run(base + [d, "one.model"], tmp_path)
    many = run(base + ["--gpus", str(ranks), "--devices", ",".join(["0"] * ranks), "--comm", "p2p", "--tune", f"allreduce_chunks={chunks}",
                       d, "many.model"], tmp_path)
    assert one.returncode == 0 and many.returncode == 0, many.stderr
    la = [l for l in one.stdout.split("\n") if l.startswith("Iter")]; lb = [l for l in many.stdout.split("\n") if l.startswith("Iter")]
    assert len(la) == len(lb) == 4
    for x, y in zip(la, lb):
        fx = [float(v) for v in re.findall(NUM, x)]; fy = [float(v) for v in re.findall(NUM, y)]
        assert np.allclose(fx[:1] + fx[2:], fy[:1] + fy[2:], rtol=2e-5, atol=2e-6), (x, y)
    a = np.frombuffer(open(tmp_path / "one.model", "rb").read(), np.float64)
    b = np.frombuffer(open(tmp_path / "many.model", "rb").read(), np.float64)
    assert a.shape == b.shape and np.nanmax(np.abs(a[2:] - b[2:])) < 1e-9 * np.nanmax(np.abs(a[2:]))


@pytest.mark.gpu
@pytest.mark.parametrize("n", [3, 5])
def test_gpus_option_three_ranks_default_precision(n, tmp_path):
    """Three / five ranks on the one GPU in the default precision (fp32 storage): quality lines within 1e-3 of one rank."""
    g, meta, d = golden_dir("mid5", tmp_path)
    base = [TRAIN, "-k", str(int(g["r"])), "-l", repr(float(g["lam"])), "-t", "3"]
    one = run(base + [d, "one.model"], tmp_path)
    three = run(base + ["--gpus", str(n), "--devices", ",".join(["0"] * n), "--comm", "p2p", d, "three.model"], tmp_path)
    assert one.returncode == 0 and three.returncode == 0, three.stderr
    pat = r"^\((Training|Testing)\) pairwise error is (\S+) and ndcg is (\S+)$"
    a = re.findall(pat, one.stdout, re.M); b = re.findall(pat, three.stdout, re.M)
    assert len(a) == len(b) == 8
    for (t1, e1, n1), (t2, e2, n2) in zip(a, b):
        assert t1 == t2 and abs(float(e1) - float(e2)) < 1e-3 and abs(float(n1) - float(n2)) < 1e-3


STUB = os.path.join(ROOT, "oracle", "_ref", "omp-pmf-train-mi355x")


def test_integration_stub_links_against_the_reference_objects(tmp_path):
    """INTEGRATION.md section 1 as a program (oracle/integration_stub.cpp): the reference's own load() / initial() /
    save_mat_t() objects, its headers, and libprimalcr.so in one binary -- built by oracle/Makefile wherever the reference is
    present (the GPU box gets the prebuilt file).  Without a GPU it must stop at pcr_solver_create with the library's error."""
    import torch
    if not os.path.exists(STUB):
        pytest.skip("oracle/_ref/omp-pmf-train-mi355x not built (no /root/reference in this container)")
    ldd = subprocess.run(["ldd", STUB], capture_output=True, text=True).stdout
    assert "libprimalcr.so" in ldd and "not found" not in ldd
    d = synth.write_dir(synth.generate("tiny"), str(tmp_path / "data"))
    r = run([STUB, "-k", "4", d], tmp_path)
    assert r.returncode == 1 and "usage:" in r.stderr
    if torch.cuda.is_available():
        pytest.skip("GPU present: the no-GPU failure path needs a machine without one")
    r = run([STUB, "-k", "4", "-t", "1", d, "m.model"], tmp_path)
    assert r.returncode == 1 and "no HIP device" in r.stderr
    assert "the number of rows is 60 and the number of cols is 40" in r.stdout      # the reference's loader ran


@pytest.mark.gpu
@pytest.mark.parametrize("solver", [2, 1])
def test_integration_stub_reproduces_the_reference_run(solver, tmp_path):
    """The reference-side binding end to end: reference loader + reference initial() + pcr_train through the C ABI + reference
    save_mat_t(), against the unmodified binary's golden stdout and model on the same directory (fp64 storage)."""
    if not os.path.exists(STUB):
        pytest.skip("oracle/_ref/omp-pmf-train-mi355x not built")
    g, meta, d = golden_dir("mid5", tmp_path)
    out = run([STUB, "-s", str(solver), "-k", str(int(g["r"])), "-l", repr(float(g["lam"])), "-t", str(meta["iters"]), "--f64", d, "stub.model"], tmp_path)
    assert out.returncode == 0, out.stderr
    strip = lambda text: [l for l in text.split("\n") if l.startswith(("Iter", "(T"))]
    ours, theirs = strip(out.stdout), strip(meta["stdout_s%d" % solver])
    assert len(ours) == len(theirs) > 0
    for a, b in zip(ours, theirs):
        fa = [float(x) for x in re.findall(NUM, a)]; fb = [float(x) for x in re.findall(NUM, b)]
        if a.startswith("Iter"):
            fa, fb = fa[:1] + fa[2:], fb[:1] + fb[2:]                # (the time differs)
        assert re.sub(NUM, "#", a) == re.sub(NUM, "#", b) and np.allclose(fa, fb, rtol=2e-5, atol=2e-6), (a, b)
    raw = open(tmp_path / "stub.model", "rb").read()                  # written by the reference's save_mat_t
    assert len(raw) == meta[f"model_bytes_s{solver}"]
    d1, k = struct.unpack("ll", raw[:16])
    U = np.frombuffer(raw, np.float64, d1 * k, 16).reshape(d1, k)
    off = 16 + 8 * d1 * k
    d2, k2 = struct.unpack("ll", raw[off:off + 16])
    V = np.frombuffer(raw, np.float64, d2 * k2, off + 16).reshape(d2, k2)
    assert np.abs(U - g[f"cli_U_s{solver}"]).max() < 1e-6 * np.abs(U).max()
    assert np.abs(V - g[f"cli_V_s{solver}"]).max() < 1e-6 * np.abs(V).max()


@pytest.mark.gpu
def test_gpus_option_large_vectors_take_the_two_phase_exchange(tmp_path):
    """V-side vectors beyond 256 KB go through the reduce-scatter + all-gather form of the peer-to-peer all-reduce (smaller
    ones through the one-shot form the golden-set tests exercise): 2500 items x k = 24 in fp64 = 480 KB per vector, three
    ranks on the one GPU (slices of unequal length), against the single-process run."""
    R = synth.generate("small", seed=5, d1=400, d2=2500, nnz=30000, mu=4.0, sigma=0.7)
    d = synth.write_dir(R, str(tmp_path / "data"))
    base = [TRAIN, "-k", "24", "-l", "50", "-t", "2", "-p", "0", "--f64"]
    one = run(base + [d, "one.model"], tmp_path)
    three = run(base + ["--gpus", "3", "--devices", "0,0,0", "--comm", "p2p", "--tune", "p2p_ll=0", d, "three.model"], tmp_path)
    assert one.returncode == 0 and three.returncode == 0, three.stderr
    strip = lambda out: [re.sub(r"time \S+", "time T", l) for l in out.split("\n") if l.startswith("Iter")]
    la, lb = strip(one.stdout), strip(three.stdout)
    assert len(la) == len(lb) == 3
    for x, y in zip(la, lb):
        assert np.allclose([float(v) for v in re.findall(NUM, x)], [float(v) for v in re.findall(NUM, y)], rtol=2e-5), (x, y)
    a = np.frombuffer(open(tmp_path / "one.model", "rb").read(), np.float64)
    b = np.frombuffer(open(tmp_path / "three.model", "rb").read(), np.float64)
    assert a.shape == b.shape and np.nanmax(np.abs(a[2:] - b[2:])) < 1e-9 * np.nanmax(np.abs(a[2:]))


@pytest.mark.gpu
def test_gpus_option_a_failing_rank_takes_the_job_down(tmp_path):
    """A device error on one half of the job must end the WHOLE job with exit code 1 -- not leave a rank waiting in a
    collective.  The fault-injection knob makes every workgroup cluster lose a member (a bounded hand-off wait -> error flag);
    the flag rides on the objective's all-reduce, so both ranks see it, report PCR_ERR_DEVICE and leave; the parent reaps them."""
    R = synth.generate("small", seed=4, d1=40, d2=6000, nnz=60000, mu=7.0, sigma=0.6)       # users of ~1000-3000 ratings
    d = synth.write_dir(R, str(tmp_path / "data"))
    cmd = [TRAIN, "-k", "8", "-t", "2", "-p", "0", "--gpus", "2", "--devices", "0,0", "--comm", "p2p", "--tune", "fault_cluster_member=1", d, "m.model"]
    r = subprocess.run(cmd, cwd=tmp_path, capture_output=True, text=True, timeout=120)
    assert r.returncode == 1, r.stderr
    assert "timed out" in r.stderr and "a GPU worker failed" in r.stderr
    ok = run([TRAIN, "-k", "8", "-t", "2", "-p", "0", "--gpus", "2", "--devices", "0,0", "--comm", "p2p", d, "m.model"], tmp_path)
    assert ok.returncode == 0, ok.stderr                               # ... and the GPU is fine afterwards


@pytest.mark.gpu
def test_gpus_option_a_rank_that_misses_an_exchange_fails_the_job_within_the_deadline(tmp_path):
    """The device-driven peer-to-peer exchange waits on a wall-clock deadline (pcr_tune p2p_timeout_ms).  Fault injection: the
    last rank never launches its 5th exchange.  Its peer's kernel must give up at the deadline (not spin for a minute), answer
    with poison instead of a made-up sum, and the job must end with exit code 1 and a time-out message -- promptly, and
    leaving the GPU usable."""
    import time
    g, meta, d = golden_dir("mid5", tmp_path)
    cmd = [TRAIN, "-k", str(int(g["r"])), "-l", repr(float(g["lam"])), "-t", "3", "-p", "0", "--gpus", "2", "--devices", "0,0", "--comm", "p2p",
           "--tune", "fault_p2p_skip=5", "--tune", "p2p_timeout_ms=700", d, "m.model"]
    t0 = time.time()
    r = subprocess.run(cmd, cwd=tmp_path, capture_output=True, text=True, timeout=120)
    took = time.time() - t0
    assert r.returncode == 1, (r.stdout, r.stderr)
    assert "timed out waiting for a peer rank" in r.stderr and "a GPU worker failed" in r.stderr
    assert os.path.getsize(tmp_path / "m.model") == 0                  # (opened before training, pmf-train.cpp:252-259) no model from a failed job
    assert took < 30, took
    ok = run([TRAIN, "-k", str(int(g["r"])), "-t", "2", "-p", "0", "--gpus", "2", "--devices", "0,0", "--comm", "p2p", d, "m.model"], tmp_path)
    assert ok.returncode == 0, ok.stderr


@pytest.mark.gpu
def test_gpus_option_without_fine_grained_boxes_every_rank_takes_the_host_path(tmp_path):
    """A rank that cannot get fine-grained memory for its exchange boxes (fault injection on the last rank) must not poll
    coarse-grained memory for stores that arrive over xGMI: it says so in the control block and ALL ranks use the
    host-synchronised exchange -- same results as the device-driven run."""
    g, meta, d = golden_dir("mid5", tmp_path)
    base = [TRAIN, "-k", str(int(g["r"])), "-l", repr(float(g["lam"])), "-t", "3", "--f64", "-p", "0", "--gpus", "2", "--devices", "0,0",
            "--comm", "p2p", "--tune", "debug=1"]
    a = run(base + [d, "a.model"], tmp_path)
    b = run(base + ["--tune", "fault_p2p_coarse=1", d, "b.model"], tmp_path)
    assert a.returncode == 0 and b.returncode == 0, b.stderr
    assert "device-driven (fine-grained boxes)" in a.stderr and "all ranks take the host-synchronised exchange" not in a.stderr
    assert "all ranks take the host-synchronised exchange" in b.stderr and "every exchange host-synchronised" in b.stderr
    la = [l for l in a.stdout.split("\n") if l.startswith("Iter")]; lb = [l for l in b.stdout.split("\n") if l.startswith("Iter")]
    assert len(la) == len(lb) == 4
    for x, y in zip(la, lb):
        fx = [float(v) for v in re.findall(NUM, x)]; fy = [float(v) for v in re.findall(NUM, y)]
        assert fx[:1] + fx[2:] == fy[:1] + fy[2:], (x, y)              # both exchanges sum in rank order: the same digits
    ma = np.frombuffer(open(tmp_path / "a.model", "rb").read(), np.float64)
    mb = np.frombuffer(open(tmp_path / "b.model", "rb").read(), np.float64)
    assert np.array_equal(ma, mb)


@pytest.mark.gpu
def test_gpus_option_snapshots_give_a_resume_point(tmp_path):
    """SURVEY 8f-4 across ranks: --snapshot-every 2 with two ranks (on the one GPU) writes <model>.iter2 / .iter4 from rows
    every rank deposits at that iteration; .iter2 equals the model of a 2-iteration run, .iter4 the final model, and 2 more
    iterations from .iter2 (--init-model, again two ranks) continue the trajectory to the 4-iteration model.
    The reference writes its model once, at the end (pmf-train.cpp:297-310)."""
    g, meta, d = golden_dir("mid5", tmp_path)
    base = [TRAIN, "-k", str(int(g["r"])), "-l", repr(float(g["lam"])), "-p", "0", "--f64", "--gpus", "2", "--devices", "0,0", "--comm", "p2p"]
    four = run(base + ["-t", "4", "--snapshot-every", "2", d, "four.model"], tmp_path)
    assert four.returncode == 0, four.stderr
    two = run(base + ["-t", "2", d, "two.model"], tmp_path)
    assert two.returncode == 0, two.stderr
    rd = lambda n: np.frombuffer(open(tmp_path / n, "rb").read(), np.float64)
    assert (tmp_path / "four.model.iter2").exists() and (tmp_path / "four.model.iter4").exists()
    assert not (tmp_path / "four.model.iter1").exists() and not (tmp_path / "four.model.iter3").exists()
    assert np.array_equal(rd("four.model.iter4"), rd("four.model"))
    assert np.array_equal(rd("four.model.iter2"), rd("two.model"))               # the exchange sums in rank order: same bits
    resumed = run(base + ["-t", "2", "--init-model", "four.model.iter2", d, "resumed.model"], tmp_path)
    assert resumed.returncode == 0, resumed.stderr
    a, b = rd("four.model"), rd("resumed.model")
    assert a.shape == b.shape and np.nanmax(np.abs(a[2:] - b[2:])) < 1e-9 * np.nanmax(np.abs(a[2:]))
    iters = lambda s: [re.sub(r"time \S+", "time T", l) for l in s.split("\n") if l.startswith("Iter ")]
    assert iters(four.stdout)[:3] == iters(two.stdout)
    # against one rank, to summation-order rounding
    one = run([TRAIN, "-k", str(int(g["r"])), "-l", repr(float(g["lam"])), "-p", "0", "--f64", "-t", "4", d, "one.model"], tmp_path)
    c = rd("one.model")
    assert one.returncode == 0 and np.nanmax(np.abs(a[2:] - c[2:])) < 1e-9 * np.nanmax(np.abs(c[2:]))


@pytest.mark.gpu
def test_gpus_option_a_signal_to_the_parent_ends_the_whole_job(tmp_path):
    """The parent of a --gpus job only waits; SIGTERM / SIGINT to it must reach the GPU workers (forward_signal), the parent must
    reap them all and exit 1 without a model -- no worker may be left behind holding the GPU."""
    import signal
    import time
    g, meta, d = golden_dir("mid5", tmp_path)
    cmd = [TRAIN, "-k", str(int(g["r"])), "-l", repr(float(g["lam"])), "-t", "200000", "-p", "0", "--gpus", "2", "--devices", "0,0", "--comm", "p2p",
           d, "m.model"]
    for sig in (signal.SIGTERM, signal.SIGINT):
        p = subprocess.Popen(cmd, cwd=tmp_path, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        time.sleep(4.0)                                               # both workers are training by now (a run of minutes)
        assert p.poll() is None, p.stderr.read()
        kids = [int(x) for x in subprocess.run(["pgrep", "-P", str(p.pid)], capture_output=True, text=True).stdout.split()]
        assert len(kids) == 2, kids
        t0 = time.time()
        p.send_signal(sig)
        out, err = p.communicate(timeout=60)
        assert p.returncode == 1, (p.returncode, err[-2000:])
        assert time.time() - t0 < 30
        assert "a GPU worker failed" in err
        assert os.path.getsize(tmp_path / "m.model") == 0
        for k in kids:                                               # reaped: neither running nor a zombie of ours
            assert not os.path.exists(f"/proc/{k}"), k
    ok = run([TRAIN, "-k", str(int(g["r"])), "-t", "2", "-p", "0", "--gpus", "2", "--devices", "0,0", "--comm", "p2p", d, "m.model"], tmp_path)
    assert ok.returncode == 0, ok.stderr


@pytest.mark.gpu
def test_gpus_option_ranks_sharing_a_device_past_its_queue_budget_take_the_host_synchronised_exchange(tmp_path):
    """NOTES.md round 6: one GPU maps 24 hardware queues at once for ALL its processes; past that, kernels of different processes
    are time-sliced instead of resident together and a hand-off between them costs 7-11 ms instead of 0.5 us (tools/ubench/
    queue_budget_probe.hip) -- the layout in which four workers on one GPU once ran into the 20 s deadline of the device-driven
    exchange.  Every rank publishes the queues it holds; four ranks with four lanes each are over the budget, so the job switches to
    the host-synchronised exchange as a whole, says so once, and still reproduces the one-process model."""
    g, meta, d = golden_dir("mid5", tmp_path)
    base = [TRAIN, "-k", str(int(g["r"])), "-l", "2.5", "-t", "2", "--f64"]
    one = run(base + [d, "one.model"], tmp_path)
    four = run(base + ["--gpus", "4", "--devices", "0,0,0,0", "--comm", "p2p", "--tune", "lanes=4", d, "four.model"], tmp_path)
    assert one.returncode == 0 and four.returncode == 0, four.stderr[-2000:]
    notes = re.findall(r"^\[pcr\] p2p: (\d+) ranks share device \S+ and hold (\d+) hardware queues", four.stderr, re.M)
    assert len(notes) == 1 and int(notes[0][0]) == 4 and int(notes[0][1]) > 16, four.stderr[-2000:]
    assert "every exchange of this job is host-synchronised" in four.stderr
    a = np.frombuffer(open(tmp_path / "one.model", "rb").read(), np.float64)
    b = np.frombuffer(open(tmp_path / "four.model", "rb").read(), np.float64)
    assert a.shape == b.shape and np.nanmax(np.abs(a[2:] - b[2:])) < 1e-9 * np.nanmax(np.abs(a[2:]))
    # within the budget (one stream per rank, what the CLI chooses by itself for ranks that share a device): no note, device-driven
    quiet = run(base + ["--gpus", "3", "--devices", "0,0,0", "--comm", "p2p", d, "three.model"], tmp_path)
    assert quiet.returncode == 0 and "[pcr] p2p:" not in quiet.stderr, quiet.stderr[-2000:]


@pytest.mark.gpu
def test_a_side_file_that_cannot_be_written_is_an_error(tmp_path):
    """pmf-train.cpp:276-295 writes U.txt / V.txt after training and ignores a failure; the drop-in CLI must not report success for a
    run whose outputs are missing: U.txt is a directory here, the run ends with a message and a non-zero exit code."""
    g, meta, d = golden_dir("edge5", tmp_path)
    (tmp_path / "U.txt").mkdir()
    r = run([TRAIN, "-k", "4", "-t", "1", "-p", "0", d, "m.model"], tmp_path)
    assert r.returncode != 0 and "U.txt" in r.stderr, (r.returncode, r.stderr[-300:])


@pytest.mark.gpu
def test_gpus_option_with_more_ranks_than_users_that_have_ratings(tmp_path):
    """One user holds nearly every rating: the nnz-balanced partition gives two of the four workers NO users.  They still take part
    in every exchange (contributing zeros) and the job's model equals the one-process model to summation-order rounding."""
    rng = np.random.default_rng(0)
    d1, d2 = 6, 300
    lens = np.array([250, 0, 3, 0, 0, 12])
    user = np.repeat(np.arange(d1), lens)
    item = np.concatenate([np.sort(rng.choice(d2, n, replace=False)) for n in lens])
    val = rng.integers(1, 6, len(user)).astype(np.float64)
    tu = np.array([0, 2, 5]); ti = np.array([1, 2, 3]); tv = np.array([3.0, 4.0, 2.0])
    d = synth.write_dir(synth.Ratings(d1, d2, user, item, val, tu, ti, tv), str(tmp_path / "data"))
    base = [TRAIN, "-k", "6", "-l", "2.5", "-t", "3", "--f64"]
    one = run(base + [d, "one.model"], tmp_path)
    four = run(base + ["--gpus", "4", "--devices", "0,0,0,0", "--comm", "p2p", d, "four.model"], tmp_path)
    assert one.returncode == 0 and four.returncode == 0, four.stderr
    shards = re.findall(r"^\[rank (\d)\] device 0: users \[(\d+), (\d+)\), (\d+) ratings$", four.stderr, re.M)
    assert len(shards) == 4 and sum(1 for _, a, b, _ in shards if a == b) >= 1          # at least one worker without users
    a = np.frombuffer(open(tmp_path / "one.model", "rb").read(), np.float64)
    b = np.frombuffer(open(tmp_path / "four.model", "rb").read(), np.float64)
    assert a.shape == b.shape and np.nanmax(np.abs(a[2:] - b[2:])) < 1e-9 * np.nanmax(np.abs(a[2:]))
    la = [l for l in one.stdout.split("\n") if l.startswith(("Iter", "(T"))]; lb = [l for l in four.stdout.split("\n") if l.startswith(("Iter", "(T"))]
    assert len(la) == len(lb) == 4 + 8
    for x, y in zip(la, lb):
        fx = [float(v) for v in re.findall(NUM, x)]; fy = [float(v) for v in re.findall(NUM, y)]
        if x.startswith("Iter"):
            fx, fy = fx[:1] + fx[2:], fy[:1] + fy[2:]
        assert np.allclose(fx, fy, rtol=2e-5, atol=2e-6, equal_nan=True), (x, y)


@pytest.mark.gpu
@pytest.mark.parametrize("args", [["-t", "0"], ["-k", "1", "-t", "2"], ["-k", "3", "-t", "2", "-s", "1"], ["-t", "1", "-l", "0.001"],
                                  ["-t", "2", "-p", "0"], ["-t", "1", "-k", "130"]],
                         ids=["no-iterations", "rank-1", "rank-3-solver-1", "tiny-lambda", "no-evaluation", "rank-130"])
def test_train_cli_corner_parameters_against_the_reference_binary(args, tmp_path):
    """Corner values of the reference's own flags, side by side with the UNMODIFIED reference binary (-n 1, deterministic) on the
    edge-case data set (users with 0 / 1 ratings, all-equal ratings, duplicate scores): the same lines to the printed digits
    (fp64 mode), the same model to 1e-7."""
    from oracle import oracle_py
    if not os.path.exists(oracle_py.REF_TRAIN):
        pytest.skip("oracle/_ref was not built (no /root/reference at build time)")
    g, meta, d = golden_dir("edge5", tmp_path)
    (tmp_path / "ref").mkdir(); (tmp_path / "ours").mkdir()
    ref = subprocess.run([oracle_py.REF_TRAIN, "-n", "1"] + args + [d, "m.model"], cwd=tmp_path / "ref", capture_output=True, text=True, timeout=300)
    ours = subprocess.run([TRAIN, "-n", "1", "--f64"] + args + [d, "m.model"], cwd=tmp_path / "ours", capture_output=True, text=True, timeout=300)
    assert ref.returncode == 0 and ours.returncode == 0, (ref.stderr[-500:], ours.stderr[-500:])
    keep = lambda out: [re.sub(r"time \S+", "time T", l) for l in out.strip().split("\n") if not l.startswith("Wall-time")]
    la, lb = keep(ref.stdout), keep(ours.stdout)
    assert len(la) == len(lb), (la, lb)
    for x, y in zip(la, lb):
        if x != y:
            assert re.sub(NUM, "#", x) == re.sub(NUM, "#", y), (x, y)
            # (users whose training ratings are all equal are driven to rounding noise by PrimalCR++; the evaluator then ranks their
            # TEST items by the sign of that noise -- DESIGN 3.8 -- so the test-set columns agree to a few of the 177 pairs only)
            tol = 4e-2 if x.startswith("(Testing)") else 2e-6
            assert np.allclose([float(v) for v in re.findall(NUM, x)], [float(v) for v in re.findall(NUM, y)], rtol=2e-5, atol=tol, equal_nan=True), (x, y)
    a = np.frombuffer(open(tmp_path / "ref" / "m.model", "rb").read(), np.float64)
    b = np.frombuffer(open(tmp_path / "ours" / "m.model", "rb").read(), np.float64)
    assert a.shape == b.shape and np.array_equal(a[:2].view(np.int64), b[:2].view(np.int64))
    assert np.nanmax(np.abs(a[2:] - b[2:])) <= 1e-7 * max(1.0, np.nanmax(np.abs(a[2:])))


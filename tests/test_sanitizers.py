"""sanitizers, on the CPU build only (GPU AddressSanitizer / XNACK runs are not available on this pool).

* `make -C primalcr_amd/csrc asan`: the host side of the product -- loader, parser, cache, partitioner, model file, knob table
  (pcr_host.cpp), both drop-in CLIs incl. the fork / shared-memory / reaping code of --gpus N, the data generator -- built with
  g++ -fsanitize=address,undefined; the [device] entry points resolve to "no HIP device" errors (sanitize/device_absent.cpp:
  no computation).  The host-side test files then run in a child pytest with that build loaded (tests/conftest.py,
  PCR_SANITIZED_DIR), and the CLIs run once more with LeakSanitizer on.
* `make -C primalcr_amd/csrc tsan`: the lock-free shared-memory rendezvous / generation barrier / error flag of the
  peer-to-peer communicator (pcr_p2p.h) under ThreadSanitizer, ranks as threads, HIP mocked (sanitize/p2p_tsan_harness.cpp).
  (It found one race: ranks read the control block's creation time before acquiring its magic word.)
The reference's own race is the shared `obj_u_new` at pcrpp.cpp:822-832."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

CSRC = os.path.join(ROOT, "primalcr_amd", "csrc")
SAN = os.path.join(ROOT, "build_san", "asan")
pytestmark = pytest.mark.skipif(bool(os.environ.get("PCR_SANITIZED_DIR")), reason="already inside the sanitized child run")


def _libasan():
    p = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    return p if os.path.isabs(p) and os.path.exists(p) else None


def _env(**extra):
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(extra)
    return env


@pytest.fixture(scope="module")
def asan_build():
    if _libasan() is None:
        pytest.skip("no libasan in this toolchain")
    subprocess.run(["make", "-s", "-C", CSRC, "asan"], check=True)
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "asan", f"SANOUT={SAN}"], check=True, stderr=subprocess.DEVNULL)
    return SAN


@pytest.mark.timeout(900)
def test_host_side_test_files_pass_under_asan_and_ubsan(asan_build):
    """tests/test_host_abi.py (ABI, loader, convert, model file, parser, cache, partitioner, knob table), the CPU half of
    tests/test_cli.py (usage, exit codes, the --gpus fork / reap path with failing workers) and tests/test_oracle_golden.py
    (the C restatement against the reference's vectors) with every native library an ASan + UBSan build: any heap error or
    undefined behaviour aborts the child run."""
    env = _env(PCR_SANITIZED_DIR=asan_build, LD_PRELOAD=_libasan(),
               ASAN_OPTIONS="detect_leaks=0:halt_on_error=1:abort_on_error=0:exitcode=86:allocator_may_return_null=1",
               UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "not gpu", "-p", "no:cacheprovider",
                        "tests/test_host_abi.py", "tests/test_cli.py", "tests/test_oracle_golden.py"],
                       cwd=ROOT, env=env, capture_output=True, text=True)
    tail = r.stdout[-3000:] + r.stderr[-3000:]
    assert r.returncode == 0, tail
    assert "AddressSanitizer" not in tail and "runtime error:" not in tail, tail
    assert " passed" in r.stdout and "failed" not in r.stdout.split("\n")[-2], tail


@pytest.mark.timeout(300)
def test_cli_host_paths_are_leak_free(asan_build, tmp_path):
    """The CLIs as stand-alone executables with LeakSanitizer ON: argument errors, a real data directory parsed and cached, the
    --gpus 3 parent that forks, reaps its (device-less) workers and takes the job down.  Exit codes are the product's own
    (1), never the sanitizer's (86)."""
    sys.path.insert(0, ROOT)
    from primalcr_amd import synth
    R = synth.generate("tiny")
    d = synth.write_dir(R, str(tmp_path / "data"))
    env = _env(ASAN_OPTIONS="detect_leaks=1:halt_on_error=1:exitcode=86", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    train, predict = os.path.join(asan_build, "omp-pmf-train"), os.path.join(asan_build, "omp-pmf-predict")
    cases = [([train], 1), ([train, "-z", "3", "dir"], 1), ([train, "-s", "7", "dir"], 0), ([train, str(tmp_path / "missing"), "m.model"], 1),
             ([train, "-k", "4", "-t", "1", "--cache", str(tmp_path / "c.bin"), d, "m.model"], 1),       # parses + caches, then: no device
             ([train, "-k", "4", "-t", "1", "--cache", str(tmp_path / "c.bin"), d, "m.model"], 1),       # ... from the cache
             ([train, "--gpus", "3", "-p", "0", d, "m.model"], 1), ([train, "--gpus", "8", "-p", "0", d, "m.model"], 1),
             ([train, "--gpus", "16", "--comm", "p2p", "-p", "0", d, "m.model"], 1), ([train, "--gpus", "17", d, "m.model"], 1), ([train, "--tune", "nonsense=1", d, "m.model"], 1),
             ([predict], 1), ([predict, "nope", "m", "o"], 1)]
    for cmd, want in cases:
        r = subprocess.run(cmd, cwd=tmp_path, env=env, capture_output=True, text=True, timeout=120)
        msg = (cmd[1:], r.returncode, r.stderr[-1500:])
        if "LeakSanitizer has encountered a fatal error" in r.stderr or "LeakSanitizer does not work under ptrace" in r.stderr:
            pytest.skip("LeakSanitizer cannot run in this container")
        assert r.returncode == want, msg
        assert "Sanitizer" not in r.stderr and "runtime error:" not in r.stderr, msg
    assert os.path.exists(tmp_path / "c.bin")


@pytest.mark.timeout(300)
def test_p2p_rendezvous_and_barriers_under_tsan():
    subprocess.run(["make", "-s", "-C", CSRC, "tsan"], check=True)
    r = subprocess.run([os.path.join(ROOT, "build_san", "tsan", "p2p_harness")], capture_output=True, text=True, timeout=240,
                       env=_env(TSAN_OPTIONS="halt_on_error=0:exitcode=66"))
    if "FATAL: ThreadSanitizer" in r.stderr and "unexpected memory mapping" in r.stderr:
        pytest.skip("ThreadSanitizer cannot map its shadow in this container")
    assert r.returncode == 0, r.stderr[-4000:]
    assert "WARNING: ThreadSanitizer" not in r.stderr, r.stderr[-4000:]
    assert "all scenarios behaved" in r.stderr
    assert "ranks 4: 4 ok, 0 failed" in r.stderr and "ranks 3: 0 ok, 3 failed" in r.stderr and "ranks 3: 0 ok, 2 failed" in r.stderr
    # the target machine's 8 ranks and the control block's 16 (no GPU box lets 8 processes share its one card: NOTES.md, round 5)
    assert "ranks 8: 8 ok, 0 failed" in r.stderr and "ranks 16: 16 ok, 0 failed" in r.stderr


@pytest.mark.timeout(300)
def test_host_data_path_threads_under_tsan():
    """The multi-threaded loader (pieces, ordered-file adoption, bucketed counting sort, running-maximum test rows), the level builder
    and the parallel pcr_initial of pcr_host.cpp under ThreadSanitizer: every stage hands its workers disjoint ranges and joins them
    before the next stage reads."""
    subprocess.run(["make", "-s", "-C", CSRC, "tsan"], check=True)
    r = subprocess.run([os.path.join(ROOT, "build_san", "tsan", "host_harness")], capture_output=True, text=True, timeout=240,
                       env=_env(TSAN_OPTIONS="halt_on_error=0:exitcode=66"))
    if "FATAL: ThreadSanitizer" in r.stderr and "unexpected memory mapping" in r.stderr:
        pytest.skip("ThreadSanitizer cannot map its shadow in this container")
    assert r.returncode == 0, r.stderr[-4000:]
    assert "WARNING: ThreadSanitizer" not in r.stderr, r.stderr[-4000:]
    assert "loader, CSR build, levels and initial() behaved" in r.stderr

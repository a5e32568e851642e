import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
# tests/test_sanitizers.py re-runs the host-side test files in a child pytest against the AddressSanitizer / UBSan build
# (make -C primalcr_amd/csrc asan): this variable (test infrastructure only -- the product reads no environment variable)
# names the directory that holds that build's libprimalcr.so, CLIs, generator and oracle
SANITIZED_DIR = os.environ.get("PCR_SANITIZED_DIR")
BIN_DIR = SANITIZED_DIR or os.path.join(ROOT, "primalcr_amd", "bin")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """A clean checkout has no built artefacts (they are git-ignored): compile the product and the checker once, exactly as
    __graft_entry__.build() does (an up-to-date tree is a no-op of make).  The product itself never builds or falls back on
    its own: without the library it fails loudly (tests/test_host_abi.py)."""
    if SANITIZED_DIR:
        import primalcr_amd as pcr
        from oracle import oracle_py
        from primalcr_amd import synth
        pcr.use_library(os.path.join(SANITIZED_DIR, "libprimalcr.so"))
        synth.use_library(os.path.join(SANITIZED_DIR, "libpcrsynth.so"))
        oracle_py.ORACLE_SO = os.path.join(SANITIZED_DIR, "libpcroracle.so")
        return
    need = [os.path.join(ROOT, "primalcr_amd", "lib", "libprimalcr.so"), os.path.join(ROOT, "primalcr_amd", "bin", "omp-pmf-train"),
            os.path.join(ROOT, "primalcr_amd", "bin", "omp-pmf-predict")]
    if not all(os.path.exists(p) for p in need):
        import __graft_entry__
        __graft_entry__.build()


@pytest.fixture(scope="session")
def oracle():
    """The plain-C CPU restatement (oracle/pcr_oracle.c) -- the checker."""
    from oracle.oracle_py import Oracle
    return Oracle()


def load_golden(name):
    import json
    g = dict(np.load(os.path.join(GOLDEN, name + ".npz")))
    with open(os.path.join(GOLDEN, name + ".json")) as f:
        meta = json.load(f)
    return g, meta


def golden_csr(g, test=False):
    from oracle.oracle_py import CSR
    p = "tcsr_" if test else "csr_"
    return CSR(int(g["d1"]), int(g["d2"]), g[p + "idx"], g[p + "item"], g[p + "val"])


GOLDEN_CASES = ["edge5", "real", "mid5"]

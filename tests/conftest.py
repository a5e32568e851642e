import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


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

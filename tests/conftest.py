import os
import sys

import numpy as np
import pytest
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


def coo_from(g, prefix):
    shape = tuple(int(s) for s in g[prefix + "_shape"])
    return sp.coo_matrix((g[prefix + "_data"], (g[prefix + "_row"], g[prefix + "_col"])),
                         shape=shape).tocsr()


@pytest.fixture(scope="session")
def golden():
    return load_golden

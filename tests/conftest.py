import json
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


def load_golden(name):
    """npz fixture -> (dict of arrays, list of case descriptors)"""
    z = np.load(os.path.join(GOLDEN, name), allow_pickle=False)
    meta = json.loads(bytes(z["meta"]).decode())
    return z, meta


@pytest.fixture(scope="session")
def oracle():
    from oracle import pyoracle
    return pyoracle.Oracle()


@pytest.fixture(scope="session")
def reference():
    from oracle import pyoracle
    if not pyoracle.have_reference():
        pytest.skip("oracle/_ref not built (needs /root/reference; see oracle/Makefile)")
    return pyoracle.Reference()


@pytest.fixture(scope="session")
def gpu():
    """The product library on cuda:0.  Fails loudly (no CPU fallback) if it cannot load."""
    import torch
    assert torch.cuda.is_available(), "GPU test selected but no GPU is visible"
    import x265_amd
    x265_amd.init(0)
    return x265_amd

import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


@pytest.fixture(scope="session")
def canon_np():
    from bin_amd.weights import canonical_weights
    return canonical_weights(0)


@pytest.fixture(scope="session")
def canon_cpu(canon_np):
    import torch
    return {k: torch.from_numpy(v) for k, v in canon_np.items()}


@pytest.fixture(scope="session")
def canon_gpu(canon_np):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return {k: torch.from_numpy(v).cuda() for k, v in canon_np.items()}


@pytest.fixture(scope="session", autouse=True)
def _built_library():
    """The HIP library must exist for every test session (CPU tests check its symbols too)."""
    from bin_amd.build import build_library
    build_library(verbose=False)

import os
import sys
import warnings

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
warnings.filterwarnings("ignore", category=UserWarning)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return dict(np.load(os.path.join(GOLDEN, name), allow_pickle=False))
    return load


def torch_state(np_state):
    import torch
    return {k: torch.from_numpy(np.asarray(v)) for k, v in np_state.items()}


@pytest.fixture(scope="session", autouse=True)
def _engine_library():
    """Build libcnm_engine.so once if the tree has not been built yet (hipcc cross-compiles without a GPU)."""
    from cnmnet_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        from cnmnet_amd.build import build
        build(verbose=False)
    yield

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


def normal_parity(got, want, ref_cam, disp_oracle, k_size=9, trust=5e-4):
    """North-star bar for normals as a statement about EVERY pixel.  The reference inverts the 81-point
    normal equations in fp32, so its own output moves by up to 3.5e-3 with the summation order; the engine sums in fp64.
    With n64 = the same fit evaluated in float64 on the oracle's depth:
      e_fit  = max |engine - n64| over ALL pixels                                   (asserted < 1e-3 by the callers)
      e_ref  = max |engine - reference| over the pixels where the reference itself is within `trust` of n64
      excluded = the fraction of pixels outside that set (where the reference's fp32 solve is the one that is off).
    got / want [B,3,H,W] torch (cpu), ref_cam [B,2,4,4], disp_oracle [B,1,H,W].  Returns (e_fit, e_ref, excluded, q99)."""
    from oracle import ref_arrangement as ra
    n64, _ = ra.depth_to_normal(1.0 / disp_oracle.double().squeeze(1), ref_cam[:, 1, :3, :3].double().inverse(), k_size)
    ref_off = (want.double() - n64).abs().amax(1)
    trusted = ref_off < trust
    d = (got.double() - want.double()).abs().amax(1)
    e_fit = float((got.double() - n64).abs().max())
    e_ref = float(d[trusted].max()) if bool(trusted.any()) else 0.0
    return e_fit, e_ref, float(1.0 - trusted.double().mean()), float(d.flatten().quantile(0.99))


def _cpu_quota():
    """CPUs this process may actually use: min(affinity, cgroup v2 cpu.max quota)."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


@pytest.fixture(scope="session", autouse=True)
def _oracle_threads():
    """The CPU oracle (torch ops) sizes its thread pool by the logical CPU count; the GPU boxes show 256 CPUs under a quota of
    16, and 256 runnable threads on 16 CPUs' worth of time made one 640x480 oracle frame take 7 minutes instead of 20 s."""
    try:
        import torch
        torch.set_num_threads(max(1, min(torch.get_num_threads(), _cpu_quota())))
    except ImportError:
        pass
    yield


@pytest.fixture(scope="session", autouse=True)
def _engine_library():
    """Build libcnm_engine.so once if the tree has not been built yet (hipcc cross-compiles without a GPU)."""
    from cnmnet_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        from cnmnet_amd.build import build
        build(verbose=False)
    yield

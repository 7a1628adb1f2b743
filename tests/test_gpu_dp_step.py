"""GPU (-m gpu): SURVEY 8 a-10 -- the data-parallel training step on the ENGINE nets.  Two ranks (both on GPU 0, gloo;
the driver's multi-GPU run uses RCCL and one GPU per rank) each take half of a B=4 batch through `TrainStepWoNormal`;
rank 0 compares gradients and parameters after the Adam step with one process running the same two half batches
(reference train.py:111-115, :164-175, :307-310).  Assertions live in tests/dp_step_worker.py."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_rank_engine_train_step():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "dp_step_worker.py")]
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=1500, env=dict(os.environ, PYTHONFAULTHANDLER="1", TORCH_SHOW_CPP_STACKTRACES="1"))   # no retry: a rank lost to a signal fails the test, with its stack
    errs = "\n".join(l for l in (r.stderr or "").splitlines() if any(w in l for w in ("Error", "assert", "what()", "terminate", "fault", "File \"")))
    assert r.returncode == 0, (r.stdout[-1500:], errs[-3000:], r.stderr[-1500:])
    assert "dp-step OK" in r.stdout and r.stdout.count("dp-graph OK") == 2, r.stdout[-1500:]

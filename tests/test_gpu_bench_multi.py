"""GPU (-m gpu): bench.py's N > 1 path end to end on a one-GPU box -- two ranks launched by torch.distributed.run, both
on GPU 0, gloo for the barrier and the max-over-ranks time (the driver's multi-GPU run uses RCCL and one GPU per rank)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_rank_bench_line():
    env = dict(os.environ, CNM_BENCH_BACKEND="gloo", CNM_BENCH_DEVICE="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29517", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--no-roofline"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                              # rank 0 prints the one line
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["scaling"] == "weak" and d["unit"] == "frames/s"
    assert abs(d["value"] - 2 * 8 * 3 / (d["ms_per_step"] * 3e-3)) < 1e-6 * d["value"]     # all ranks' frames / slowest rank's time
    assert "cpu_baseline" not in d                                         # N = 1 only

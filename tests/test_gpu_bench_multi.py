"""GPU (-m gpu): bench.py's N > 1 path end to end on a one-GPU box -- two ranks launched by torch.distributed.run, both
on GPU 0, gloo for the barrier and the max-over-ranks time (the driver's multi-GPU run uses RCCL and one GPU per rank)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _launch(cmd, env, timeout, amd_log=False):
    """Run a multi-rank bench command, ONCE (no retry: a rank lost to a signal fails the test).  [r6] Every launch is logged UNCONDITIONALLY:
    per-rank stderr (faulthandler, TORCH_SHOW_CPP_STACKTRACES; AMD_LOG_LEVEL=2 for the eight-rank runs) and per-rank resource lines go to
    gpurun_out/multi_rank_logs/<launch>/ from the first instruction on, the parent empties its caching allocator first and records what it
    still holds on the GPU the ranks are about to share.  On success the directory is removed and its resource lines are appended to
    gpurun_out/multi_rank_resources.txt (the record of the runs that did NOT fail); on failure everything stays."""
    import shutil
    import time
    import torch
    tag = "%s_%d_%d" % (os.environ.get("PYTEST_CURRENT_TEST", "launch").split("::")[-1].split(" ")[0], os.getpid(), int(time.time() * 1000) % 10**9)
    d = os.path.join(ROOT, "gpurun_out", "multi_rank_logs", tag)
    os.makedirs(d, exist_ok=True)
    parent = "parent pid %d: " % os.getpid()
    try:
        if torch.cuda.is_available() and torch.cuda.is_initialized():
            torch.cuda.synchronize(); torch.cuda.empty_cache()
            free, total = torch.cuda.mem_get_info()
            parent += "device free %.2f of %.2f GB after empty_cache (reserved %.2f GB, allocated %.2f GB), " % (
                free / 2**30, total / 2**30, torch.cuda.memory_reserved() / 2**30, torch.cuda.memory_allocated() / 2**30)
        parent += "open fds %d" % len(os.listdir("/proc/self/fd"))
    except Exception as e:                                               # noqa: BLE001
        parent += "(%s)" % e
    with open(os.path.join(d, "parent.txt"), "w") as f:
        f.write(parent + "\n" + " ".join(cmd) + "\n")
    env = dict(env, PYTHONFAULTHANDLER="1", TORCH_SHOW_CPP_STACKTRACES="1", CNM_RANK_LOG_DIR=d)
    if amd_log:
        env["AMD_LOG_LEVEL"] = "2"
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    with open(os.path.join(d, "launcher.txt"), "w") as f:
        f.write("---- exit code %d\n---- stdout\n%s\n---- stderr\n%s\n" % (r.returncode, r.stdout, r.stderr))
    if r.returncode == 0:
        try:
            with open(os.path.join(ROOT, "gpurun_out", "multi_rank_resources.txt"), "a") as out:
                out.write("==== %s ok\n%s\n" % (tag, parent))
                for name in sorted(os.listdir(d)):
                    if name.startswith("rank") and name.endswith(".txt"):
                        out.write(open(os.path.join(d, name)).read())
            shutil.rmtree(d, ignore_errors=True)
        except OSError:
            pass
    else:                                                                 # what the ranks wrote to their own stderr files, for the assertion message
        extra = []
        for name in sorted(os.listdir(d)):
            if name.endswith(".err"):
                extra.append("---- %s\n%s" % (name, open(os.path.join(d, name), errors="replace").read()[-4000:]))
        r.stderr = (r.stderr or "") + "\n" + "\n".join(extra)
    return r


def _check_line(r, world=2, frames=8):
    assert r.returncode == 0, _why(r)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                              # rank 0 prints the one line
    d = json.loads(lines[0])
    assert d["n_gpus"] == world and d["steps"] == 3 and d["scaling"] == "weak" and d["unit"] == "frames/s"
    assert abs(d["value"] - world * frames * 3 / (d["ms_per_step"] * 3e-3)) < 1e-6 * d["value"]     # all ranks' frames / slowest rank's time
    assert "cpu_baseline" not in d and "f16" not in d                      # N = 1 only
    assert d["step_ms"]["n"] == 3 and d["step_ms"]["p10"] <= d["step_ms"]["median"] <= d["step_ms"]["p90"]
    assert d["per_rank_frames_per_s"]["min"] <= d["per_rank_frames_per_s"]["max"]
    assert abs(d["barrier_bracketed_wall_s"] - d["ms_per_step"] * 3e-3) < 1e-9


def test_two_rank_bench_line():
    """Started the way the driver starts N > 1: under torch.distributed.run."""
    env = dict(os.environ, CNM_BENCH_BACKEND="gloo", CNM_BENCH_DEVICE="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29517", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--no-roofline"]
    _check_line(_launch(cmd, env, 600))


def test_two_rank_bench_self_launch():
    """Started the way the driver starts N = 1: plain `python bench.py --gpus 2` spawns its own ranks."""
    env = dict(os.environ, CNM_BENCH_BACKEND="gloo", CNM_BENCH_DEVICE="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--no-roofline"]
    _check_line(_launch(cmd, env, 600))


def test_two_rank_train_mode_line():
    """bench.py --mode train (BASELINE configs[2]): two ranks on GPU 0 over gloo take the `train` step on a shard each and
    exchange gradients with the bucketed all-reduce; the line carries samples/s and the exchange's accounting."""
    env = dict(os.environ, CNM_BENCH_BACKEND="gloo", CNM_BENCH_DEVICE="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--mode", "train", "--gpus", "2", "--steps", "2", "--warmup", "1", "--samples-per-gpu", "1"]
    r = _launch(cmd, env, 900)
    assert r.returncode == 0, _why(r)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["unit"] == "samples/s" and d["n_gpus"] == 2 and d["config"]["global_batch"] == 2 and d["config"]["parallelism"] == "dp2"
    assert abs(d["value"] - 2 * 1 * 2 / (d["ms_per_step"] * 2e-3)) < 1e-6 * d["value"]
    ar = d["allreduce"]
    assert ar["bytes_per_step"] == 4 * 44674566 and ar["buckets"] >= 3                    # 33 898 500 + 10 776 066 parameters
    assert ar["launched_from_backward_hooks"] + ar["launched_late"] == ar["buckets"] and ar["exposed_ms"] > 0


def _why(r):
    """The ranks' own error lines (the launcher's summary at the end of stderr names only the signal)."""
    err = r.stderr or ""
    keep = [l for l in err.splitlines() if any(w in l for w in ("Error", "error", "what()", "terminate", "Abort", "abort", "fault", "assert", "Traceback", "File \"")) and "amdgpu.ids" not in l]
    return "\n".join(keep[:60]) + "\n...\n" + err[-1500:]


def _clean_env():
    env = dict(os.environ, CNM_BENCH_BACKEND="gloo", CNM_BENCH_DEVICE="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "OMP_NUM_THREADS"):
        env.pop(k, None)
    return env


def test_eight_rank_dry_run_eval():
    """the driver's 8-GPU form, dry-run on ONE GPU -- eight ranks self-launched by `python bench.py --gpus 8`
    (rendezvous on 127.0.0.1, LOCAL_RANK -> device mapping overridden to GPU 0, gloo for the barrier and the timing reduce,
    host threads capped per rank), one frame per rank: the line has the N = 8 shape and the whole-job value."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "3", "--warmup", "1", "--frames-per-gpu", "1",
           "--no-roofline", "--no-secondary"]
    r = _launch(cmd, _clean_env(), 1200, amd_log=True)
    _check_line(r, world=8, frames=1)
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert d["config"]["barrier_backend"] == "gloo" and d["config"]["host_threads_per_rank"] >= 1


def test_eight_rank_dry_run_train():
    """The same for --mode train (BASELINE configs[2]: 8 ranks, data parallel): every bucket of the gradient exchange leaves
    from a backward hook on every step (the overlapped path), the line carries the global batch of 8 x 1."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--mode", "train", "--gpus", "8", "--steps", "2", "--warmup", "1", "--samples-per-gpu", "1"]
    r = _launch(cmd, _clean_env(), 1800, amd_log=True)
    assert r.returncode == 0, _why(r)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["unit"] == "samples/s" and d["n_gpus"] == 8 and d["config"]["global_batch"] == 8 and d["config"]["parallelism"] == "dp8"
    assert abs(d["value"] - 8 * 1 * 2 / (d["ms_per_step"] * 2e-3)) < 1e-6 * d["value"]
    ar = d["allreduce"]
    assert ar["bytes_per_step"] == 4 * 44674566 and ar["launched_from_backward_hooks"] == ar["buckets"] and ar["launched_late"] == 0


def test_two_rank_f16_line():
    """BASELINE configs[4] ("fp16 path ... batch=16 on 2xMI355X"): two ranks of 8 frames each through the f16 engine, dry-run on one GPU."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--precision", "f16", "--gpus", "2", "--steps", "3", "--warmup", "1", "--no-roofline", "--no-secondary"]
    r = _launch(cmd, _clean_env(), 900)
    _check_line(r, world=2, frames=8)
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert d["dtype"].startswith("f16") and d["config"]["frames_per_gpu"] == 8


def test_bench_graph_mode_with_roofline():
    """`python bench.py --graph` with the roofline on (the path that died of an UnboundLocalError): one GPU, the step replayed as
    a HIP graph, the line carries the roofline objects and the store policy the graph was captured with."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--graph", "--steps", "3", "--warmup", "1", "--no-secondary", "--no-cpu-baseline", "--no-live-traffic"]
    env = _clean_env()
    env.pop("CNM_BENCH_BACKEND", None); env.pop("CNM_BENCH_DEVICE", None)
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, _why(r)
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["roofline"]["bound"] == "mfma" and d["roofline_planesweep"]["bound"] == "hbm"
    assert d["roofline_planesweep"]["store_policy"]["in_force"] in ("plain", "nt") and "Graph" in d["config"]["launch"]

"""CPU, world_size-2 gloo: the N>1 path of bench.py is host logic only (independent frame shards,
barrier, max-over-ranks timing) -- exercised here without a GPU."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from cnmnet_amd import sharding


def test_shard_range_partitions_exactly():
    for total in (0, 1, 7, 8, 9, 64):
        for world in (1, 2, 3, 8):
            spans = [sharding.shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [e - s for s, e in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        sharding.shard_range(8, 2, 2)


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lo, hi = sharding.shard_range(9, rank, world)
        local = 1.0 + rank                      # rank 1 is the slow one
        dist.barrier()
        elapsed = sharding.job_elapsed(local, dist)
        thr = sharding.job_throughput(hi - lo, local, dist)
        out.put((rank, lo, hi, elapsed, thr))
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_max_time_and_sum_units():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get() for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert [(r[1], r[2]) for r in res] == [(0, 5), (5, 9)]
    assert all(abs(r[3] - 2.0) < 1e-12 for r in res)            # max over ranks
    assert all(abs(r[4] - 9 / 2.0) < 1e-12 for r in res)        # all units / slowest rank


def test_single_process_is_identity():
    assert sharding.job_elapsed(0.25) == 0.25 and sharding.job_throughput(8, 0.5) == 16.0


def _ddp_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from cnmnet_amd.trainer import BucketedGradAllReduce
        torch.manual_seed(0)
        net = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.ReLU(), torch.nn.Linear(16, 4), torch.nn.Linear(4, 1))
        red = BucketedGradAllReduce(net.parameters(), dist, bucket_bytes=256)       # several buckets
        x = torch.arange(16, dtype=torch.float32).view(2, 8) * (rank + 1) / 10
        for _ in range(2):                                                          # two steps: state resets between steps
            net.zero_grad()
            (net(x).sum() + net(x * 0.5).sum()).backward()                          # two forwards, one backward
            red.finish()
        out.put((rank, len(red.buckets), [p.grad.clone() for p in net.parameters()]))
    finally:
        dist.destroy_process_group()


def test_bucketed_grad_allreduce_two_ranks_gloo():
    """The training path's only exchange step: bucketed gradient all-reduce == mean of per-rank grads."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    procs = [ctx.Process(target=_ddp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get() for _ in procs), key=lambda r: r[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert res[0][1] >= 2
    for a, b in zip(res[0][2], res[1][2]):
        assert torch.equal(a, b)                                                    # identical averaged gradients on both ranks
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.ReLU(), torch.nn.Linear(16, 4), torch.nn.Linear(4, 1))
    want = None
    for rank in range(2):
        net.zero_grad()
        x = torch.arange(16, dtype=torch.float32).view(2, 8) * (rank + 1) / 10
        (net(x).sum() + net(x * 0.5).sum()).backward()
        g = [p.grad.clone() for p in net.parameters()]
        want = g if want is None else [u + v for u, v in zip(want, g)]
    for a, w in zip(res[0][2], want):
        assert torch.allclose(a, w / 2, atol=1e-6)


def _unused_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from cnmnet_amd.trainer import BucketedGradAllReduce
        torch.manual_seed(0)
        used = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.ReLU(), torch.nn.Linear(16, 1))
        idle = torch.nn.Linear(8, 3)                                                # a branch the loss does not touch
        params = list(idle.parameters()) + list(used.parameters())
        opt = torch.optim.Adam(params, lr=1e-2, weight_decay=1e-1)
        red = BucketedGradAllReduce(params, dist, bucket_bytes=64)
        before = [p.detach().clone() for p in idle.parameters()]
        x = torch.arange(16, dtype=torch.float32).view(2, 8) * (rank + 1) / 10
        for _ in range(3):
            opt.zero_grad(set_to_none=False)
            used(x).sum().backward()
            red.finish()
            opt.step()
        ok = all(p.grad is None for p in idle.parameters()) and all(torch.equal(a, p.detach()) for a, p in zip(before, idle.parameters()))
        ok = ok and all((p not in opt.state or len(opt.state[p]) == 0) for p in idle.parameters())
        out.put((rank, ok, red.late_launches, [p.detach().numpy().tolist() for p in used.parameters()]))
    finally:
        dist.destroy_process_group()


def test_reducer_leaves_unused_parameters_alone_gloo():
    """Parameters without a gradient (the probability decoder during warm-up epochs) keep grad None on every rank:
    Adam with weight decay must not touch them (it would decay them towards zero), exactly as on one device."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    procs = [ctx.Process(target=_unused_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get() for _ in procs), key=lambda r: r[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert res[0][1] and res[1][1]
    assert res[0][2] >= 1                                                           # the idle bucket left from finish(), not from a hook
    assert res[0][3] == res[1][3]                                                   # the used branch stays in lock step


def _attach_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from cnmnet_amd.trainer import BucketedGradAllReduce
        torch.manual_seed(0)
        used = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.ReLU(), torch.nn.Linear(16, 1))
        idle = torch.nn.Linear(8, 3)
        params = list(idle.parameters()) + list(used.parameters())
        red = BucketedGradAllReduce(params, dist, bucket_bytes=64)
        x = torch.arange(16, dtype=torch.float32).view(2, 8) * (rank + 1) / 10
        in_place = True
        for _ in range(3):
            red.attach()                                                            # [r6] gradients accumulate straight into the buckets
            (used(x).sum() + used(0.5 * x).sum()).backward()
            in_place = in_place and all(p.grad.data_ptr() == red.view_of[id(p)].data_ptr() for p in used.parameters())
            red.finish()
            in_place = in_place and all(p.grad.data_ptr() == red.view_of[id(p)].data_ptr() for p in used.parameters())
        out.put((rank, in_place, all(p.grad is None for p in idle.parameters()), red.hook_launches, red.late_launches, len(red.buckets),
                 [p.grad.clone() for p in used.parameters()]))
    finally:
        dist.destroy_process_group()


def test_gradients_live_in_the_buckets_gloo():
    """[r6] attach(): `.grad` IS the bucket slice before backward (autograd accumulates into it), during the all-reduce and after it
    (averaged in place) -- no copy in, no copy out; a parameter without a gradient gets None back; the values equal the mean of the
    per-rank gradients."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    procs = [ctx.Process(target=_attach_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get() for _ in procs), key=lambda r: r[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for r in res:
        assert r[1] and r[2] and r[3] + r[4] == r[5] and r[4] >= 1, r[:6]
    torch.manual_seed(0)
    used = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.ReLU(), torch.nn.Linear(16, 1))
    want = None
    for rank in range(2):
        used.zero_grad()
        x = torch.arange(16, dtype=torch.float32).view(2, 8) * (rank + 1) / 10
        (used(x).sum() + used(0.5 * x).sum()).backward()
        g = [p.grad.clone() for p in used.parameters()]
        want = g if want is None else [u + v for u, v in zip(want, g)]
    for a, b, w in zip(res[0][6], res[1][6], want):
        assert torch.equal(a, b) and torch.allclose(a, w / 2, atol=1e-6)


def _init_dist_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), CNM_BENCH_FAIL_NCCL="1")
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    d, pg, backend = bench.init_dist("nccl", "cpu", world)
    try:
        t = torch.tensor([float(rank + 1)])
        d.all_reduce(t, group=pg)                                                   # the group every later barrier / all-reduce of bench.py uses
        out.put((rank, backend, pg is None, float(t.item())))
    finally:
        d.destroy_process_group()


def test_init_dist_failure_path_lands_every_rank_on_gloo():
    """bench.py's RCCL set-up failing (injected) is a tested path -- every rank agrees on gloo, the data-path group is the
    default group, collectives work, and the backend string that goes into the JSON line says so."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    procs = [ctx.Process(target=_init_dist_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get() for _ in procs)
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert [(r[1], r[2], r[3]) for r in res] == [("gloo", True, 3.0), ("gloo", True, 3.0)]

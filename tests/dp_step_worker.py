"""Worker of tests/test_gpu_dp_step.py (one process per rank, started by torch.distributed.run; both ranks on GPU 0,
gloo): a data-parallel `TrainStepWoNormal` on the engine nets, checked on rank 0 against a one-process run of the same
two half batches (reference train.py:111-115 DataParallel semantics: per-replica BatchNorm statistics, gradients
averaged over the replicas)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cnmnet_amd import synthetic as syn                              # noqa: E402
from cnmnet_amd.depthnet import depthNet, DepthRefineNet            # noqa: E402
from cnmnet_amd import trainer as trainer_mod                        # noqa: E402
from cnmnet_amd.trainer import TrainStepWoNormal, synthetic_training_sample, make_adam   # noqa: E402

H, W, B, LR = 64, 96, 4, 1e-4


def nets(dev):
    out = []
    for m, seed in ((depthNet(3.0), 61), (DepthRefineNet(32, 3.0), 62)):
        shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
        w = syn.state_dict_like(shapes, seed=seed, randomize_bn=False)
        m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in w.items()})
        out.append(m.to(dev).train())
    return out


def named(dn, rn):
    return [("d." + k, p) for k, p in dn.named_parameters()] + [("r." + k, p) for k, p in rn.named_parameters()]


def l2rel(a, b):
    return float((a - b).norm() / (b.norm() + 1e-30))


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("gloo")
    try:
        sample = synthetic_training_sample(B, H, W, seed=11, device=dev)
        keys = ("rgbs", "cameras", "disparities", "depths")
        per = B // world
        shard = {k: sample[k][rank * per:(rank + 1) * per].contiguous() for k in keys}

        # ---- data-parallel step: every bucket must leave from a backward hook (the overlapped path)
        dn, rn = nets(dev)
        init = {k: p.detach().clone() for k, p in named(dn, rn)}
        step = TrainStepWoNormal(dn, rn, lr=LR, dist=dist, exact_masked_means=True)
        assert step.reducer is not None and len(step.reducer.buckets) >= 3
        logs = step(**shard)
        assert np.isfinite(logs["loss"])
        assert step.reducer.hook_launches == len(step.reducer.buckets) and step.reducer.late_launches == 0, \
            (step.reducer.hook_launches, step.reducer.late_launches, len(step.reducer.buckets))
        dp_grad = {k: p.grad.detach().clone() for k, p in named(dn, rn)}
        dp_par = {k: p.detach().clone() for k, p in named(dn, rn)}
        # both ranks hold the same averaged gradients and the same parameters
        for k in ("d.conv1.0.weight", "d.disp1.0.bias", "r.prob.0.weight", "r.conv3.3.weight"):
            t = dp_par[k].cpu().clone()
            lst = [torch.zeros_like(t) for _ in range(world)]
            dist.all_gather(lst, t)
            assert all(torch.equal(lst[0], u) for u in lst), k

        # ---- warm-up epoch (train.py:555-559): the probability decoder gets no gradient -> untouched on every rank
        dn2, rn2 = nets(dev)
        step2 = TrainStepWoNormal(dn2, rn2, lr=LR, dist=dist, exact_masked_means=True)
        step2(**shard, warmup_epoch=True)
        unused = [(k, p) for k, p in rn2.named_parameters() if p.grad is None]
        assert unused and all(("prob" in k) for k, _ in unused), [k for k, _ in unused][:5]
        assert any(k.startswith("prob.") for k, _ in unused)
        for k, p in unused:
            assert p not in step2.optimizer.state or len(step2.optimizer.state[p]) == 0, k
            assert torch.equal(p.detach(), init["r." + k]), k
        assert step2.reducer.late_launches >= 1                              # their buckets cannot leave from a hook

        # ---- rank 0: the same two half batches in ONE process (BatchNorm statistics per half batch)
        if rank == 0:
            dn1, rn1 = nets(dev)
            one = TrainStepWoNormal(dn1, rn1, lr=LR)
            halves = [{k: sample[k][r * per:(r + 1) * per].contiguous() for k in keys} for r in range(world)]
            one.optimizer.zero_grad(set_to_none=False)
            total = sum(one.losses(**h)[0] for h in halves) / world
            total.backward()
            worst_g = worst_u = 0.0
            for k, p in named(dn1, rn1):
                e = l2rel(dp_grad[k], p.grad)
                worst_g = max(worst_g, e)
                assert e < 1e-3, ("gradient", k, e)
            one.optimizer.step()
            # Adam's FIRST update is lr * g / (|g| + eps): sign-like, so an element whose gradient is near zero may move by
            # +lr on one side and -lr on the other although the gradients agree to 1e-3 -- the update itself is only bounded
            # (|delta| <= lr on each side); what must agree are the optimizer's MOMENTS, which are linear (exp_avg = 0.1 g)
            # and quadratic (exp_avg_sq = 0.001 g^2) in the averaged gradient.
            for (k, p), (_, pd) in zip(named(dn1, rn1), named(dn, rn)):
                assert float((dp_par[k] - p.detach()).abs().max()) <= 2.0 * LR * 1.001, k
                st1, std = one.optimizer.state[p], step.optimizer.state[pd]
                e1, e2 = l2rel(std["exp_avg"], st1["exp_avg"]), l2rel(std["exp_avg_sq"], st1["exp_avg_sq"])
                worst_u = max(worst_u, e1, e2)
                assert e1 < 1e-3 and e2 < 2e-3, ("Adam moments", k, e1, e2)
                assert float(std["step"]) == float(st1["step"]) == 1.0, k
            print("dp-step OK: worst gradient rel-L2 %.2e, worst Adam-moment rel-L2 %.2e, buckets %d" % (worst_g, worst_u, len(step.reducer.buckets)), flush=True)
        # ---- graph mode next to the reducer: the step replayed as two HIP graphs (forward + refine backward | depthNet backward) with
        # the refine net's buckets launched in between, the rest and Adam eager; two steps on different shards against the eager
        # data-parallel step
        # Both forms of the forward: depthNet over the two sources in one pass (SplitSources outputs), and the reference's two calls,
        # whose outputs are NESTED (disp1 = head(iconv1), both handed on): the cut between the two graphs must be a true cut there
        # too (cutting at depthNet's own outputs double-counted the head paths / raised "backward through the graph a second time").
        for one_pass in (True, False):
            trainer_mod.SOURCES_IN_ONE_PASS = one_pass
            pair = []
            for graph in (False, True):
                dg, rg = nets(dev)
                st = TrainStepWoNormal(dg, rg, lr=LR, dist=dist, graph=graph)
                st.optimizer = make_adam(list(rg.parameters()) + list(dg.parameters()), LR, 1e-5, capturable=True)
                if graph:
                    assert st.reducer is not None and not st.reducer.handles
                for sd in (11, 12):
                    smp = synthetic_training_sample(B, H, W, seed=sd, device=dev)
                    lg = st(**{k: smp[k][rank * per:(rank + 1) * per].contiguous() for k in keys})
                    assert np.isfinite(lg["loss"])
                if graph:
                    # the segmented replay: the refine net's buckets (no bucket mixes the two nets) left between the two graphs, the
                    # depthNet buckets after the second one
                    # [r5] three graphs (refine | depthNet decoder | depthNet encoder): the refine net's AND the decoder's buckets leave early
                    r = st.reducer
                    n_refine = sum(all(id(p) in st._refine_ids for p in b) for b in r.buckets)
                    n_early = sum(all(id(p) in st._early_ids for p in b) for b in r.buckets)
                    assert st._graph_b is not None and st._graph_c is not None and n_refine >= 1 and n_early > n_refine, (n_refine, n_early)
                    assert r.hook_launches == n_early and r.hook_launches + r.late_launches == len(r.buckets), \
                        (n_refine, n_early, r.hook_launches, r.late_launches, len(r.buckets))
                pair.append((lg, {k: p.detach().clone() for k, p in named(dg, rg)}))
            (le, pe), (lgr, pg_) = pair
            assert abs(le["loss"] - lgr["loss"]) <= 1e-5 * max(1.0, abs(le["loss"])), (one_pass, le["loss"], lgr["loss"])
            worst = max(float((pe[k] - pg_[k]).abs().max()) for k in pe)
            assert worst <= 2e-6, (one_pass, worst)                                 # two Adam steps of 1e-4
            if rank == 0:
                print("dp-graph OK: eager vs graphed data-parallel step (%s), worst parameter difference %.2e" % ("both sources in one pass" if one_pass else "two depthNet calls, nested outputs", worst), flush=True)
        trainer_mod.SOURCES_IN_ONE_PASS = True
        dist.barrier()
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

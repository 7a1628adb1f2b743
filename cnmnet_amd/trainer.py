"""Training step of the reference (`train_wo_normal`, reference train.py:413-656) on the HIP engine,
data-parallel over one process per GPU.

* The step (`train.py:509-562`): two depthNet forwards (ref+src1, ref+src2; each with its own BatchNorm
  batch statistics, as in the reference), DepthRefineNet, the loss mix of `train.py:524-556`
  (the missing `ProbLoss` of `train.py:30,551` is only logged there and excluded from `loss_train`),
  `optimizer.zero_grad -> backward -> step` with Adam lr 1e-4, weight decay 1e-5 (configs/config.yaml:16-19).
* Data parallelism: the reference wraps the nets in single-process `nn.DataParallel` (`train.py:464-468`):
  scatter, replicate every step, gather, loss on GPU 0, reduce-add gradients.  Here every rank owns a
  batch shard and a full replica; the ONE exchange is the gradient all-reduce (44.67 M fp32 = 178.7 MB),
  done by `BucketedGradAllReduce`: ~25 MB buckets in reverse registration order, each launched
  asynchronously from a post-accumulate-grad hook the moment its last gradient is ready, so RCCL (backend
  "nccl" on ROCm) overlaps with the rest of backward; BatchNorm keeps per-replica statistics exactly like
  DataParallel.  `exact_masked_means=True` divides every masked loss sum by the GLOBAL mask count
  (one tiny all-reduce of counts), which reproduces DataParallel's gathered-batch masked means.
"""
import os
import torch

from .depthnet.losses import IdepthLoss_234, _valid, total, row_totals, mean_all
from .depthnet.depthNet_model import DepthRefineNet as _EngineRefine


class _step_scope:
    """Forward + backward of ONE optimisation step: packed filters are shared inside it (autograd.PACK_CACHE) and dropped at
    its end -- the weights change right after it, in optimizer.step()."""

    def __init__(self, plan=None, launch=True):
        """plan: the autograd.PackPlan of this step variant -- its 3x3 filter packs as ONE launch when the scope opens [r6]; launch=False
        (the later graphs of a segmented capture): the packs were launched by the scope of the step's first graph, only hand them out."""
        self.plan, self.launch = plan, launch

    def __enter__(self):
        from . import autograd
        autograd.PACK_CACHE = {}
        autograd.PACK_PLAN = self.plan
        if self.plan is not None:
            self.plan.run(self.launch)
        return self

    def __exit__(self, *exc):
        from . import autograd
        autograd.PACK_CACHE = None
        autograd.PACK_PLAN = None
        if self.plan is not None and exc[0] is None:
            self.plan.freeze()
        return False


# _zero_grad_note: on one GPU the eager step drops the gradients (set_to_none=True) instead of zeroing them -- backward then ASSIGNS
# every gradient where it would otherwise launch one `grad += new` per parameter (217 launches, 1.1 ms of GPU time per step), which is
# also what the captured step does.  Same update for every parameter that receives a gradient; a parameter that receives none (the
# probability decoder during the warm-up epochs, train.py:555-559) is skipped by Adam either way until its first gradient.  With a
# gradient reducer the buffers stay allocated (its buckets copy from / to them).
FUSED_MASKED_L1 = os.environ.get("CNM_FUSED_MASKED_L1", "1") != "0"   # the masked mean-L1 loss terms as one HIP launch each way (autograd.MaskedL1)
FUSED_NORMAL_TERMS = os.environ.get("CNM_FUSED_NORMAL_TERMS", "1") != "0"   # [r6] the surface-normal loss terms as two HIP launches forward, one backward (autograd.NormalCosTerms)
FUSED_ADAM = os.environ.get("CNM_FUSED_ADAM", "1") != "0"   # torch's single-kernel Adam on GPU parameters (the multi-tensor form otherwise)


def make_adam(params, lr=1e-4, weight_decay=1e-5, capturable=False):
    """torch.optim.Adam as the reference configures it (utils/misc.py:31-33), in the cheapest launch form: fused on the GPU
    (the whole update in a few multi-tensor kernels instead of nine foreach passes over the 217 parameters: ~1.0 -> ~0.3 ms
    of the step), foreach otherwise.  Either form makes zero_grad(set_to_none=False) one multi-tensor launch."""
    params = list(params)
    if FUSED_ADAM and params and all(p.is_cuda for p in params):
        return torch.optim.Adam(params, lr=lr, weight_decay=weight_decay, capturable=capturable, fused=True)
    return torch.optim.Adam(params, lr=lr, weight_decay=weight_decay, capturable=capturable, foreach=True)


SOURCES_IN_ONE_PASS = os.environ.get("CNM_SOURCES_IN_ONE_PASS", "1") != "0"   # depthNet over both sources of a frame in one pass with per-source BatchNorm statistics (depthNet.forward_sources)


ENCODER_BLOCKS = ("conv1", "conv2", "conv3", "conv4", "conv5")      # depthNet's contracting half (depthNet_model.py:134-141)
GRAPH_CUTS = int(os.environ.get("CNM_GRAPH_CUTS", "2"))             # multi-rank graph step: 2 = three graphs (refine | decoder | encoder), 1 = two (refine | depthNet): A/B


class BucketedGradAllReduce:
    """Average gradients across ranks with bucketed, backward-overlapped all-reduces.

    A parameter that receives no gradient in a step (e.g. the whole probability decoder during the warm-up epochs of
    `train_wo_normal`, reference train.py:555-559) contributes zeros to its bucket -- every rank builds the same
    graph, so the unused set is the same everywhere and the collective shapes match -- and its `.grad` is left
    exactly as it was (None stays None): the optimizer then skips it as it does on one device and under the
    reference's DataParallel, instead of decaying it towards zero.

    [r6] Gradients LIVE in the buckets: `views[i][k]` is parameter k's slice of `flat[i]`, shaped like the parameter.  The eager
    step attaches the views as `.grad` before backward (`attach()`: one fill per bucket instead of one per parameter), autograd
    accumulates straight into the bucket, and a bucket leaves as it is -- no copy-in; after the exchange one division per bucket
    averages in place and `.grad` keeps pointing at the view -- no copy-out (rounds 3-5 copied 2 x 179 MB per step through ~480
    small launches).  The HIP-graph step, whose gradients are tensors of the graph's memory pool, copies them into the views INSIDE
    the captured graphs (`capture_copy`), so a replay is followed by nothing but the collectives."""

    def __init__(self, params, dist, bucket_bytes=25 * 2**20, group=None, hooks=True, segment_of=None):
        """group: the process group of the all-reduces (None = the default group); hooks=False: no backward hooks -- the
        caller reduces buckets with launch_ready() / reduce_all() (the HIP-graph step: hooks do not run in a replay).
        segment_of: optional {id(parameter): segment}; a bucket never mixes segments, so that the buckets of a segment can
        leave as soon as that segment's part of backward has run (the segmented graph step: refine net, then depthNet)."""
        self.dist, self.group, self.world = dist, group, dist.get_world_size(group)
        self.params = [p for p in params if p.requires_grad]
        self.buckets, cur, size = [], [], 0
        seg = (lambda p: segment_of.get(id(p))) if segment_of else (lambda p: None)
        for p in reversed(self.params):                    # decoder grads are ready first
            if cur and seg(p) != seg(cur[-1]):
                self.buckets.append(cur); cur, size = [], 0
            cur.append(p); size += p.numel() * 4
            if size >= bucket_bytes:
                self.buckets.append(cur); cur, size = [], 0
        if cur:
            self.buckets.append(cur)
        self.bucket_of = {id(p): i for i, b in enumerate(self.buckets) for p in b}
        self.flat = [torch.zeros(sum(p.numel() for p in b), device=b[0].device, dtype=torch.float32) for b in self.buckets]
        self.views = []
        for b, f in zip(self.buckets, self.flat):
            off, vs = 0, []
            for p in b:
                vs.append(f[off:off + p.numel()].view_as(p)); off += p.numel()
            self.views.append(vs)
        self.view_of = {id(p): v for b, vs in zip(self.buckets, self.views) for p, v in zip(b, vs)}
        self.pending, self.work = [len(b) for b in self.buckets], [None] * len(self.buckets)
        self.fired = set()                                 # ids of the parameters whose gradient arrived this step
        self.hook_launches = self.late_launches = 0        # buckets launched from backward hooks / from finish(), last step
        self._hook_count = 0
        self.bytes_per_step = 4 * sum(f.numel() for f in self.flat)         # payload of one step's all-reduces
        self.handles = [p.register_post_accumulate_grad_hook(self._hook) for p in self.params] if hooks else []

    def _hook(self, p):
        i = self.bucket_of[id(p)]
        self.fired.add(id(p))
        self.pending[i] -= 1
        if self.pending[i] == 0:
            self._hook_count += 1
            self._launch(i)

    def attach(self):
        """Eager step, instead of optimizer.zero_grad(): zero the buckets and make every parameter's `.grad` its bucket view, so that
        backward accumulates straight into the buckets.  A parameter that then receives no gradient gets `.grad = None` back in finish()."""
        for f in self.flat:
            f.zero_()
        for b, vs in zip(self.buckets, self.views):
            for p, v in zip(b, vs):
                p.grad = v

    def capture_copy(self, params):
        """Inside a stream capture: copy the gradients the captured backward has just produced (tensors of the graph's pool) into their
        bucket views, so that a replay leaves the buckets ready for the collective."""
        for p in params:
            v = self.view_of.get(id(p))
            if v is not None and p.grad is not None and p.grad.data_ptr() != v.data_ptr():
                v.copy_(p.grad)

    def _launch(self, i):
        for p, v in zip(self.buckets[i], self.views[i]):
            g = p.grad
            if id(p) in self.fired and g is not None and g.data_ptr() != v.data_ptr():
                v.copy_(g)                                 # a gradient autograd ASSIGNED instead of accumulating into the view (the step right after a capture; callers that never attach())
            elif id(p) not in self.fired and (g is None or g.data_ptr() != v.data_ptr()):
                v.zero_()                                  # no gradient this step and the slice may hold an older one
        self.work[i] = self.dist.all_reduce(self.flat[i], op=self.dist.ReduceOp.SUM, group=self.group, async_op=True)

    def launch_ready(self, ready):
        """Without hooks: launch every bucket whose parameters all satisfy ready(p) and that has not left yet (the part of
        backward that produces them has been enqueued); returns the number launched.  They count as early launches."""
        n = 0
        for i, b in enumerate(self.buckets):
            if self.work[i] is None and all(ready(p) for p in b):
                for p in b:
                    if p.grad is not None:
                        self.fired.add(id(p))
                self._hook_count += 1
                self._launch(i)
                n += 1
        return n

    def reduce_all(self):
        """Without hooks: every parameter that has a gradient counts as arrived; launches all buckets, then finish()."""
        for p in self.params:
            if p.grad is not None:
                self.fired.add(id(p))
        self.finish()

    def finish(self):
        """Call after backward(): waits for all buckets and writes the averaged gradients back."""
        late = 0
        for i, b in enumerate(self.buckets):
            if self.work[i] is None:                       # a bucket holding parameters without a gradient this step
                late += 1
                self._launch(i)
        for i, b in enumerate(self.buckets):
            self.work[i].wait()
            self.flat[i].div_(self.world)                  # the average, in place: one launch per bucket
            for p, v in zip(b, self.views[i]):
                if id(p) in self.fired:
                    p.grad = v                             # the averaged gradient IS the bucket slice: no copy back
                elif p.grad is not None and p.grad.data_ptr() == v.data_ptr():
                    p.grad = None                          # attached by attach(), never written: as on one device, the optimizer skips it
            self.work[i], self.pending[i] = None, len(b)
        self.hook_launches, self.late_launches, self._hook_count = self._hook_count, late, 0
        self.fired.clear()

    def remove(self):
        for h in self.handles:
            h.remove()


def _masked_l1(pred, gt, dist=None, weight=None, exact=False, group=None):
    """IdepthLoss / IdepthwithProbLoss (reference losses.py:30-73), optionally normalised by the global
    mask count so that averaging the per-rank gradients equals the gathered-batch loss.
    `group`: the process group of the gradient exchange (the trainer's `group`): the mask count is reduced over the SAME
    ranks the gradients are averaged over, on the same backend (with bench.py's init_dist the default group is gloo and the
    gradient group RCCL: the default group would move a device scalar through the host in every loss term).
    Static shapes: the reference gathers `pred[mask]` (a device-to-host synchronisation per loss term, which stops the
    host from enqueueing the backward pass while the forward pass still runs); here masked-out elements are replaced by
    zeros BEFORE the difference (so a non-finite ground truth never reaches the arithmetic or the gradient) and the sum
    is divided by the mask count -- the same mean, NaN for an empty mask (0 / 0) as the reference's mean of nothing."""
    multi = exact and dist is not None and dist.is_initialized() and dist.get_world_size(group) > 1
    if FUSED_MASKED_L1 and pred.is_cuda and pred.dtype == torch.float32 and not multi:
        from .autograd import MaskedL1
        return MaskedL1.apply(pred, gt.expand_as(pred), weight.expand_as(pred) if weight is not None else None)   # one launch each way
    m = _valid(pred, gt)
    zero = torch.zeros((), dtype=pred.dtype, device=pred.device)
    diff = (torch.where(m, pred, zero) - torch.where(m, gt, zero)).abs()
    if weight is not None:
        diff = diff * torch.where(m, weight, zero)
    n = total(m).to(pred.dtype)                      # `total`: reductions without scratch memory (HIP-graph replay, see losses.total)
    if multi:
        n = n.reshape(1).clone()
        dist.all_reduce(n, group=group)
        return total(diff) / (n[0] / dist.get_world_size(group)).clamp(min=1.0)
    return total(diff) / n


def _log_values(logs):
    """dict of scalar tensors -> dict of floats with ONE device-to-host copy."""
    keys = list(logs)
    vals = torch.stack([logs[k].detach().float().reshape(()) for k in keys]).tolist()
    # the copy above waited for the step: if a stream-K hand-off of a staged kernel timed out inside it, say so here -- in a graph
    # replay no entry point runs that could refuse, this is the only place the failure surfaces (include/cnm_engine.h, cnm_engine_status)
    if vals and logs[keys[0]].is_cuda:
        from . import ops
        ops.engine_status(clear=False)
    return dict(zip(keys, vals))


class TrainStepWoNormal:
    """One optimisation step of `train_wo_normal` (reference train.py:509-562).

    graph=True (one GPU, no collective): forward, backward and the Adam update are captured ONCE per input shape into a
    HIP graph and replayed -- the step's ~2900 kernel launches and ~700 small torch operations no longer pass through
    Python, which is what bounds the eager step (the GPU waits for the host in the low-resolution layers).  Same
    arithmetic as the eager step: every loss term has static shapes (`_masked_l1`), the capture's warm-up iterations are
    undone in place (parameters, BatchNorm statistics, Adam state) before the first replay."""

    def __init__(self, depth_net, refine_net, lr=1e-4, weight_decay=1e-5, dist=None, exact_masked_means=False, graph=False, group=None):
        """group: process group of the gradient all-reduce (None = default).  graph=True with more than one rank: the step is
        replayed as TWO HIP graphs -- forward + the refine net's backward, then depthNet's backward -- with the refine net's
        gradient buckets handed to the collective in between (a collective stays outside a captured region, and the backward
        hooks that overlap buckets with backward do not exist in a replay); the remaining buckets and Adam follow eagerly."""
        self.depth_net, self.refine_net, self.dist, self.exact, self.group = depth_net, refine_net, dist, exact_masked_means, group
        params = list(refine_net.parameters()) + list(depth_net.parameters())              # train.py:87, :446
        self.optimizer = make_adam(params, lr, weight_decay, capturable=bool(graph))          # utils/misc.py:31-33
        self.reducer = None
        if dist is not None and dist.is_initialized() and dist.get_world_size(group) > 1:
            # segments of the segmented graph step (see _capture): refine net | depthNet's decoder | depthNet's encoder
            seg = {id(p): 0 for p in refine_net.parameters()}
            seg.update({id(p): (2 if n.split(".")[0] in ENCODER_BLOCKS else 1) for n, p in depth_net.named_parameters()})
            self.reducer = BucketedGradAllReduce(params, dist, group=group, hooks=not graph, segment_of=seg if graph else None)
        self.l234 = IdepthLoss_234()
        self.graph_mode, self._graph, self._graph_b, self._graph_c, self._graph_key, self._cut = bool(graph), None, None, None, None, None
        self._pack_plans = {}                                            # step variant -> autograd.PackPlan (the step's 3x3 filter packs as one launch)
        self.finish_events = None                                        # set to [] to collect (start, end) HIP events around the reducer's finish()

    def __call__(self, rgbs, cameras, disparities, depths, warmup_epoch=False):
        """rgbs [B,3,3,H,W] (ref, src1, src2), cameras [B,3,2,4,4], disparities / depths [B,V,1,H,W]
        (ground truth of the reference view at index 0).  Returns a dict of detached scalars."""
        if self.graph_mode:
            return self._graphed_step((rgbs, cameras, disparities, depths), bool(warmup_epoch),
                                      lambda *a: self.losses(*a, warmup_epoch))
        with _step_scope(self._plan((tuple(rgbs.shape), bool(warmup_epoch)))):
            loss, logs = self.losses(rgbs, cameras, disparities, depths, warmup_epoch)
            if self.reducer is None:
                self.optimizer.zero_grad(set_to_none=True)                                   # :562-565 (see _zero_grad_note)
            else:
                self.reducer.attach()                                                        # gradients accumulate straight into the all-reduce buckets
            loss.backward()
        if self.reducer is not None:
            self._finish(self.reducer.finish)
        self.optimizer.step()
        return _log_values(logs)

    def _plan(self, key):
        from . import autograd
        p = self._pack_plans.get(key)
        if p is None:
            p = self._pack_plans[key] = autograd.PackPlan()
        return p

    def _finish(self, fn):
        """The part of the gradient exchange that backward did not hide: buckets still in flight + the write-back."""
        if self.finish_events is None:
            return fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record()
        self.finish_events.append((e0, e1))

    # ---- HIP-graph replay of the step
    def _graphed_step(self, inputs, variant, fn):
        """inputs: tensors copied into the graph's static buffers; fn(*static inputs) -> (loss, logs) is what gets captured."""
        key = (tuple(tuple(t.shape) for t in inputs), variant)
        if self._graph_key != key:
            self._capture(inputs, fn, key)
        for dst, src in zip(self._static_in, inputs):
            dst.copy_(src)
        self._graph.replay()
        if self.reducer is not None:
            # Two graphs (see _capture): forward + the refine net's backward, then depthNet's backward.  The refine net's buckets
            # leave between the two replays, so their exchange runs under depthNet's backward (the larger half of the step's
            # backward time); the depthNet buckets and the update follow eagerly.
            refine_ids = self._refine_ids
            self.reducer.launch_ready(lambda p: id(p) in refine_ids)
            self._graph_b.replay()
            if self._graph_c is not None:                  # [r5] third graph: the decoder's buckets leave under the encoder's backward
                early = self._early_ids
                self.reducer.launch_ready(lambda p: id(p) in early)
                self._graph_c.replay()
            self._finish(self.reducer.reduce_all)
            self.optimizer.step()
        return _log_values(self._static_logs)

    def _state_tensors(self):
        """Every tensor one step mutates: parameters, BatchNorm buffers, Adam moments and step counters."""
        out = [p.data for g in self.optimizer.param_groups for p in g["params"]]
        out += [b for net in (self.depth_net, self.refine_net) for b in net.buffers()]
        for st in self.optimizer.state.values():
            out += [v for v in st.values() if torch.is_tensor(v)]
        return out

    def _capture(self, inputs, fn, key):
        self._static_in = [t.detach().clone() for t in inputs]
        saved = [t.clone() for t in self._state_tensors()]  # parameters, BatchNorm buffers and whatever Adam state exists already
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        segmented = self.reducer is not None
        refine_params = [p for p in self.refine_net.parameters() if p.requires_grad]
        self._refine_ids = {id(p) for p in refine_params}
        three = segmented and GRAPH_CUTS >= 2 and hasattr(self.depth_net, "_forward_train")
        decoder_params = [p for n, p in self.depth_net.named_parameters() if p.requires_grad and n.split(".")[0] not in ENCODER_BLOCKS]
        self._early_ids = self._refine_ids | {id(p) for p in decoder_params}

        def backward_decoder(cut, gcut):
            """Backward from depthNet's outputs down to what its encoder handed the decoder (the identity nodes depthNet records in
            `_enc_cut`): the decoder's parameter gradients are final after it; returns what the encoder's backward starts from."""
            enc = [c for c in self.depth_net._enc_cut if c.requires_grad]
            grads = torch.autograd.grad(cut, decoder_params + enc, grad_outputs=gcut, allow_unused=True)
            for p, g in zip(decoder_params, grads):
                p.grad = g
            ge = grads[len(decoder_params):]
            return [c for c, g in zip(enc, ge) if g is not None], [g for g in ge if g is not None]

        def forward(*a):
            self.depth_net._enc_cut = [] if three else None
            return fn(*a)

        def backward_refine(loss):
            """Backward from the loss down to depthNet's outputs (the cut recorded by losses()): the refine net's parameter gradients
            are final after it; returns what depthNet's backward starts from."""
            cut = [c for c in self._cut if c.requires_grad]
            grads = torch.autograd.grad(loss, refine_params + cut, allow_unused=True)
            for p, g in zip(refine_params, grads):
                p.grad = g                                  # assigned, as the captured AccumulateGrad of a None gradient does
            gc = grads[len(refine_params):]
            return [c for c, g in zip(cut, gc) if g is not None], [g for g in gc if g is not None]

        plan = self._plan(("graph",) + tuple(key))
        with torch.cuda.stream(side):                       # warm-up off the default stream: Adam state and allocator pools exist before capture
            for _ in range(2):
                with _step_scope(plan):
                    loss, logs = forward(*self._static_in)
                    self.optimizer.zero_grad(set_to_none=True)
                    if segmented:
                        cut, gcut = backward_refine(loss)
                        if three:
                            cut, gcut = backward_decoder(cut, gcut)
                        torch.autograd.backward(cut, gcut)
                        del cut, gcut
                    else:
                        loss.backward()
                if self.reducer is None:
                    self.optimizer.step()
                del loss, logs                              # no autograd graph of the warm-up may outlive it (its AccumulateGrad nodes carry their stream)
        torch.cuda.current_stream().wait_stream(side)
        self.optimizer.zero_grad(set_to_none=True)
        self._graph, self._graph_b, self._graph_c = torch.cuda.CUDAGraph(), None, None
        if segmented:
            # graph A: forward + backward through the refine net; graph B (same memory pool): depthNet's backward -- [r5] its DECODER's,
            # with graph C for the encoder's.  Between the replays the gradient buckets of the part just finished are handed to the collective.
            depth_params = [p for p in self.depth_net.parameters() if p.requires_grad]
            with torch.cuda.graph(self._graph), _step_scope(plan):
                loss, logs = forward(*self._static_in)
                cut, gcut = backward_refine(loss)
                self.reducer.capture_copy(refine_params)                 # [r6] the gradients land in their all-reduce buckets inside the graph
            self._graph_b = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self._graph_b, pool=self._graph.pool()), _step_scope(plan, launch=False):
                if three:
                    cut, gcut = backward_decoder(cut, gcut)
                    self.reducer.capture_copy(decoder_params)
                else:
                    torch.autograd.backward(cut, gcut)
                    self.reducer.capture_copy(depth_params)
            if three:
                self._graph_c = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self._graph_c, pool=self._graph.pool()), _step_scope(plan, launch=False):
                    torch.autograd.backward(cut, gcut)
                    self.reducer.capture_copy([p for p in depth_params if id(p) not in self._early_ids])
            del cut, gcut
            self.depth_net._enc_cut = None
        else:
            with torch.cuda.graph(self._graph), _step_scope(plan):
                loss, logs = forward(*self._static_in)
                loss.backward()
                if self.reducer is None:
                    self.optimizer.step()
        self._static_logs = {k: v.detach() for k, v in logs.items()}
        self._cut = None
        del loss, logs
        # undo the warm-up in place (the graph holds these addresses): the first replay is the first real step.  State that
        # existed before is restored; state the warm-up CREATED (Adam moments and step counters of parameters that had
        # none -- all of them on the first capture, the probability decoder's when a warm-up-epoch graph is followed by the
        # full one) starts from zero, exactly as if the optimizer met those parameters for the first time.  Adam appends
        # new state entries behind the existing ones, so the saved list is a prefix of the live one.
        live = self._state_tensors()
        for t, s0 in zip(live, saved):
            t.copy_(s0)
        for t in live[len(saved):]:
            t.zero_()
        self._graph_key = key

    def _cut_here(self, p01, f01, p02, f02):
        """A TRUE cut between depthNet and everything behind it: every tensor depthNet hands on goes through an
        identity autograd node (`view_as`), the refine net and the loss terms see only those, and the nodes are what the
        segmented graph step differentiates to and restarts from.  depthNet's own outputs are nested -- disp1 = head(iconv1),
        iconv1 is handed on as well -- so cutting at THEM would make `autograd.grad(loss, refine_params + cut)` walk through
        depthNet's decoder (freeing its saved tensors) and the second backward count the head paths twice; the identity
        nodes sit strictly behind depthNet and are ancestors of nothing but the refine net and the losses."""
        cut = []

        def node(t):
            v = t.view_as(t)
            cut.append(v)
            return v

        def feature(f):
            c4 = getattr(f, "_cnm_c4", None)
            if c4 is None:
                return node(f)
            c = node(c4)
            if isinstance(self.refine_net, _EngineRefine):
                g = f.detach()                               # the engine's DepthRefineNet reads the c4 tensor: the NCHW twin only carries it
            else:
                from . import autograd as _ag                # any other refine net reads the NCHW tensor: rebuilt from the cut node, so both forms
                g = _ag.C4ToNCHW.apply(c, f.shape[1])        # carry autograd through the cut (the detached twin silently lost this gradient)
            g._cnm_c4 = c
            return g

        out = ([node(t) for t in p01], feature(f01), [node(t) for t in p02], feature(f02))
        self._cut = cut
        return out

    def losses(self, rgbs, cameras, disparities, depths, warmup_epoch=False):
        """Train-mode forward of both nets on one batch (shard) and the loss mix of train.py:522-559:
        (loss to back-propagate, dict of logged terms)."""
        self.depth_net.train(); self.refine_net.train()
        gt_id, gt_d = disparities[:, 0], depths[:, 0]
        if SOURCES_IN_ONE_PASS and hasattr(self.depth_net, "forward_sources"):
            # the two depthNet calls of the reference as ONE pass over 2 B pairs whose BatchNorms keep per-source batch statistics:
            # same outputs, running statistics and gradients (tests/test_gpu_training.py), convolutions at twice the batch
            (p01, f01), (p02, f02) = self.depth_net.forward_sources(rgbs[:, 0], rgbs[:, 1:3], cameras[:, 0], cameras[:, 1:3])
        else:
            p01, f01 = self.depth_net(rgbs[:, 0], rgbs[:, 1], cameras[:, 0], cameras[:, 1])     # :509-512
            p02, f02 = self.depth_net(rgbs[:, 0], rgbs[:, 2], cameras[:, 0], cameras[:, 2])
        p01, f01, p02, f02 = self._cut_here(p01, f01, p02, f02)               # what depthNet hands on: the segmented graph step cuts backward here
        idr, prob = self.refine_net(idepth01=p01[0], idepth02=p02[0], iconv01=f01, iconv02=f02)   # :517-520
        L = lambda a, b, w=None: _masked_l1(a, b, self.dist, w, self.exact, self.group)
        loss_idepth_1 = (L(p01[0], gt_id) + L(p02[0], gt_id)) * 0.5                          # :522-523
        loss_idepth_refined = L(idr, gt_id)                                                  # :525
        loss_idepth_234 = (self.l234(p01, gt_id) + self.l234(p02, gt_id)) * 0.5              # :527-528
        eps = 1e-8
        d01, d02, dr = 1.0 / (p01[0] + eps), 1.0 / (p02[0] + eps), 1.0 / (idr + eps)          # :530-534
        prob_loss_depth = L(idr, gt_id, prob) + L(dr, gt_d, prob)                            # :539-540
        prob_loss = 5 * prob_loss_depth + (1 - mean_all(prob))                                  # :541, :545
        loss_depth_1 = (L(d01, gt_d) + L(d02, gt_d)) * 0.5                                    # :550-551
        loss_depth_refined = L(dr, gt_d)                                                     # :553
        if warmup_epoch:                                                                     # :555-559
            loss = loss_idepth_1 + loss_idepth_234 + loss_idepth_refined
        else:
            loss = loss_depth_1 + loss_depth_refined + (loss_idepth_1 + loss_idepth_234 + loss_idepth_refined) + prob_loss
        return loss, {"loss": loss, "loss_idepth": loss_idepth_1, "loss_idepth_refined": loss_idepth_refined,
                      "loss_depth_refined": loss_depth_refined, "prob_loss": prob_loss}


def synthetic_training_sample(B, H, W, seed=0, device="cpu"):
    """Seeded sample with the keys/shapes `train.py:489-507` reads (3 views): smooth ground-truth depth in
    [0.6, 4] m for the reference view, disparities = 1/depth, a few invalid (zero) pixels."""
    import numpy as np
    from . import synthetic as syn
    img, cams = syn.frames(B, 2, H, W, seed=seed)
    rng = np.random.default_rng(seed + 1)
    ys, xs = np.mgrid[0:H, 0:W].astype(np.float32)
    depth = np.stack([1.5 + 0.8 * np.sin(xs / (20 + 3 * b)) * np.cos(ys / 17.0) + 0.5 * rng.random() + 0.002 * xs for b in range(B)])
    depth = np.clip(depth, 0.6, 4.0).astype(np.float32)
    depth[:, :4, :6] = 0.0                                                  # holes: masked by the losses
    disp = np.where(depth > 0, 1.0 / np.maximum(depth, 1e-6), 0.0).astype(np.float32)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
    d = t(depth)[:, None, None].expand(B, 3, 1, H, W).contiguous()
    i = t(disp)[:, None, None].expand(B, 3, 1, H, W).contiguous()
    # ground-truth normals of the reference view from the ground-truth depth (finite differences of the point cloud)
    K = syn.intrinsics(H, W)
    X = (xs - K[0, 2]) / K[0, 0] * depth; Y = (ys - K[1, 2]) / K[1, 1] * depth
    P = np.stack([X, Y, depth], 1)                                          # [B,3,H,W]
    du = np.gradient(P, axis=3); dv = np.gradient(P, axis=2)
    n = np.cross(dv, du, axis=1); n /= (np.linalg.norm(n, axis=1, keepdims=True) + 1e-8)
    n = t(n.astype(np.float32))[:, None].expand(B, 3, 3, H, W).contiguous()
    return {"rgbs": t(img), "cameras": t(cams), "depths": d, "disparities": i, "normals": n}


def get_warped_depth_loss(depth_refined, gt_depth_src, pose, intrinsic, intrinsic_inv, inverse_warp=None):
    """Stand-in for the reference's MISSING `fusion_depth.fuse_depth.get_warped_depth_loss`
    (imported at train.py:34, called at :287-293; not in the repository -- parity unpinned).
    Definition chosen here: sample the source view's ground-truth depth at the
    re-projection of the refined reference depth (`inverse_warp`, gradient w.r.t. the refined depth through the
    sampling position) and take the masked L1 against the refined depth."""
    if inverse_warp is None:
        from .depthnet.inverse_warp import inverse_warp
    warped = inverse_warp(gt_depth_src.unsqueeze(1), depth_refined, pose, intrinsic, intrinsic_inv).squeeze(1)
    if FUSED_MASKED_L1 and warped.is_cuda and warped.dtype == torch.float32 and depth_refined.dtype == torch.float32:
        from .autograd import MaskedL1Both
        return MaskedL1Both.apply(warped, depth_refined)                       # [r6] one launch each way instead of ~25 torch launches per view
    m = (warped > 0) & torch.isfinite(warped) & torch.isfinite(depth_refined) & (depth_refined > 0)
    zero = torch.zeros((), dtype=warped.dtype, device=warped.device)
    diff = (torch.where(m, warped, zero) - torch.where(m, depth_refined, zero)).abs()
    return total(diff) / total(m).to(warped.dtype).clamp(min=1.0)          # empty mask: 0 with a zero gradient; no host synchronisation


class TrainStep(TrainStepWoNormal):
    """One optimisation step of the reference's `train` command (train.py:164-310): the depth / inverse-depth /
    probability terms of `train_wo_normal` plus surface-normal losses on normals derived from the three predicted
    depth maps (Depth2normal, k_size 9, gradient through the least-squares fit) and two warped-depth losses.
    Not restated: the plane-instance inputs the shipped loaders never produce (train.py:147-162) --
    i.e. the `use_normal_refined_by_planes=False` branch (:226-241) is the one built."""

    def __init__(self, depth_net, refine_net, k_size=9, depth2normal=None, inverse_warp=None, intrinsics_inverse=None, **kw):
        """The three geometry operators default to the engine's (HIP kernels with their backward passes); a test can
        hand in CPU restatements to evaluate the same loss on the oracle's nets."""
        super().__init__(depth_net, refine_net, **kw)
        if depth2normal is None:
            from .depthnet.depth_util import Depth2normal
            depth2normal = Depth2normal(k_size)
        if intrinsics_inverse is None:
            from . import ops
            intrinsics_inverse = ops.intrinsics_inverse
        self.depth2normal, self.inverse_warp, self.intrinsics_inverse = depth2normal, inverse_warp, intrinsics_inverse

    def __call__(self, rgbs, cameras, disparities, depths, normals):
        if self.graph_mode:                                  # the relative poses (a 4x4 inverse) are computed outside the captured region
            return self._graphed_step((rgbs, cameras, disparities, depths, normals, self.relative_poses(cameras)), "normals",
                                      lambda r, c, i, d, n, poses: self.losses(r, c, i, d, n, poses))
        with _step_scope(self._plan((tuple(rgbs.shape), "normals"))):
            loss, logs = self.losses(rgbs, cameras, disparities, depths, normals)
            if self.reducer is None:
                self.optimizer.zero_grad(set_to_none=True)                                   # :307-310 (see _zero_grad_note)
            else:
                self.reducer.attach()
            loss.backward()
        if self.reducer is not None:
            self._finish(self.reducer.finish)
        self.optimizer.step()
        return _log_values(logs)

    @staticmethod
    def relative_poses(cameras):
        """[B,2,3,4]: reference camera -> source view v = 1, 2 (train.py:284-286); inv_ex: no device-to-host error check."""
        ref_inv = torch.linalg.inv_ex(cameras[:, 0, 0], check_errors=False)[0]
        return torch.stack([(cameras[:, v, 0] @ ref_inv)[:, :3, :] for v in (1, 2)], 1).contiguous()

    @staticmethod
    def _normal_terms(pred, gt, valid):
        """Per-sample (sum of 1 - cos over the kept pixels, kept-pixel count) of `surface_normal_loss` (losses.py:76-122)."""
        if FUSED_NORMAL_TERMS and pred.is_cuda and pred.dtype == torch.float32 and gt.dtype == torch.float32:
            from .autograd import NormalCosTerms
            return NormalCosTerms.apply(pred, gt.expand_as(pred), valid.expand(pred.shape[0], 1, pred.shape[2], pred.shape[3]))   # two launches forward, one backward
        finite = torch.isfinite(gt.sum(1, keepdim=True)) & torch.isfinite(pred.sum(1, keepdim=True))
        keep = finite & valid
        zero = torch.zeros((), dtype=pred.dtype, device=pred.device)
        sim = torch.nn.functional.cosine_similarity(torch.where(keep, pred, zero), torch.where(keep, gt, zero), dim=1)
        k = keep.squeeze(1).to(sim.dtype)
        return row_totals((1 - sim) * k), row_totals(k)

    def losses(self, rgbs, cameras, disparities, depths, normals, poses=None):
        """Train-mode forward and the loss mix of train.py:164-304: (loss to back-propagate, dict of logged terms).
        Static shapes and no host decision: the reference's NaN guard (:275-280 -- a per-sample normal loss is NaN exactly
        when a sample keeps no pixel, a mean over nothing) becomes a 0 / 1 factor on the device, with the per-sample means
        computed on clamped counts so that the dropped terms contribute exact zeros to the gradients as well."""
        self.depth_net.train(); self.refine_net.train()
        gt_id, gt_d, gt_n = disparities[:, 0], depths[:, 0], normals[:, 0]
        gt_n_valid = gt_d > 0.1                                                              # :153
        if SOURCES_IN_ONE_PASS and hasattr(self.depth_net, "forward_sources"):           # as in TrainStepWoNormal.losses
            (p01, f01), (p02, f02) = self.depth_net.forward_sources(rgbs[:, 0], rgbs[:, 1:3], cameras[:, 0], cameras[:, 1:3])
        else:
            p01, f01 = self.depth_net(rgbs[:, 0], rgbs[:, 1], cameras[:, 0], cameras[:, 1])     # :164-167
            p02, f02 = self.depth_net(rgbs[:, 0], rgbs[:, 2], cameras[:, 0], cameras[:, 2])
        p01, f01, p02, f02 = self._cut_here(p01, f01, p02, f02)               # see TrainStepWoNormal.losses
        idr, prob = self.refine_net(idepth01=p01[0], idepth02=p02[0], iconv01=f01, iconv02=f02)   # :172-175
        L = lambda a, b, w=None: _masked_l1(a, b, self.dist, w, self.exact, self.group)
        loss_idepth_1 = (L(p01[0], gt_id) + L(p02[0], gt_id)) * 0.5                          # :177-178
        loss_idepth_refined = L(idr, gt_id)                                                  # :180
        d01, d02 = 1.0 / p01[0].squeeze(1), 1.0 / p02[0].squeeze(1)                          # :185-186
        dr = 1.0 / (idr.squeeze(1) + 1e-5)                                                   # :188
        prob_loss = 5 * (L(idr, gt_id, prob) + L(dr.unsqueeze(1), gt_d, prob)) + (1 - mean_all(prob))   # :193-199
        k_inv = self.intrinsics_inverse(cameras[:, 0])                                       # :201-202
        n01, _ = self.depth2normal(d01, k_inv)                                               # :204-207
        n02, _ = self.depth2normal(d02, k_inv)
        nr, _ = self.depth2normal(dr, k_inv)
        loss_depth_1 = (L(d01.unsqueeze(1), gt_d) + L(d02.unsqueeze(1), gt_d)) * 0.5          # :217-218
        loss_depth_refined = L(dr.unsqueeze(1), gt_d)                                        # :220
        (s1, c1), (s2, c2), (sr, cr) = (self._normal_terms(n, gt_n, gt_n_valid) for n in (n01, n02, nr))   # :226-263, per sample
        ln = ((s1 / c1.clamp(min=1.0) + s2 / c2.clamp(min=1.0)) * 0.5).mean()                # :268-269
        lnr = (sr / cr.clamp(min=1.0)).mean()
        ok = (torch.minimum(torch.minimum(c1, c2), cr) > 0).all()                            # :275-280: no per-sample mean over nothing
        zero = torch.zeros((), dtype=ln.dtype, device=ln.device)
        loss = loss_idepth_1 + loss_depth_1 + loss_depth_refined + loss_idepth_refined + torch.where(ok, ln + lnr + prob_loss, zero)
        K = cameras[:, 0, 1, :3, :3].contiguous()
        if poses is None:
            poses = self.relative_poses(cameras)
        for v in (0, 1):                                                                     # :284-293, :304
            loss = loss + get_warped_depth_loss(dr, depths[:, v + 1, 0], poses[:, v].contiguous(), K, k_inv, self.inverse_warp)
        nan = torch.full((), float("nan"), dtype=ln.dtype, device=ln.device)
        return loss, {"loss": loss, "loss_normal": torch.where(ok, ln, nan), "loss_normal_refined": torch.where(ok, lnr, nan),
                      "loss_depth_refined": loss_depth_refined, "prob_loss": prob_loss}


# ------------------------------------------------------------------ epoch loop and checkpoints (train.py:59-140, :395-410)
def checkpoint_name(epoch, idepth_scale):
    return "network_epoch_%d_scale_%d.pt" % (epoch, int(idepth_scale))                      # train.py:410


def save_checkpoint(path, step, epoch, global_step):
    """The reference's checkpoint dictionary (train.py:403-409): un-prefixed state_dicts of both nets + optimizer."""
    torch.save({"epoch": epoch, "global_step": global_step,
                "depth_network_state_dict": step.depth_net.state_dict(),
                "depth_refine_network_state_dict": step.refine_net.state_dict(),
                "optimizer": step.optimizer.state_dict()}, path)


def resume(path, step, with_optimizer=False):
    """train.py:92-106: load both nets (a checkpoint without the refine net is accepted), return (start_epoch, global_step).
    The reference leaves the optimizer state behind (`:103` commented out); `with_optimizer=True` restores it."""
    from .eval7scenes import load_checkpoint
    ck = torch.load(path, map_location="cpu")
    load_checkpoint(ck, step.depth_net, step.refine_net if "depth_refine_network_state_dict" in ck else None)
    if with_optimizer and "optimizer" in ck:
        step.optimizer.load_state_dict(ck["optimizer"])
    return int(ck["epoch"]), int(ck["global_step"])


def fit(step, loader, num_epochs, checkpoint_dir=None, idepth_scale=3.0, start_epoch=0, global_step=0, device="cuda:0",
        print_interval=10, log=print, rank=0, max_steps=None):
    """The reference's main loop: epochs start_epoch+1 .. num_epochs-1 (train.py:140), the first four epochs after (re)start are the
    inverse-depth-only warm-up where the step has one (`(epoch - start_epoch) < 5`, train.py:556-560), a checkpoint every len(loader)//8 iterations
    (:401-410; rank 0 only).  `step` is a TrainStep / TrainStepWoNormal; batches come from scannet.training_loader."""
    import os
    import time
    every = max(1, len(loader) // 8)
    for epoch in range(start_epoch + 1, num_epochs):
        tic = time.time()
        for it, batch in enumerate(loader):
            s = {k: v.to(device, non_blocking=True) for k, v in batch.items() if torch.is_tensor(v)}
            if isinstance(step, TrainStep):
                out = step(s["rgbs"], s["cameras"], s["disparities"], s["depths"], s["normals"])
            else:
                out = step(s["rgbs"], s["cameras"], s["disparities"], s["depths"], warmup_epoch=(epoch - start_epoch) < 5)
            global_step += 1
            if rank == 0 and it % print_interval == 0:
                log("epoch %d iter %d/%d  %s  %.3f s/iter" % (epoch, it, len(loader), "  ".join("%s %.4f" % kv for kv in out.items()),
                                                               (time.time() - tic) / (it + 1)))
            if rank == 0 and checkpoint_dir is not None and it % every == 0:
                os.makedirs(checkpoint_dir, exist_ok=True)
                save_checkpoint(os.path.join(checkpoint_dir, checkpoint_name(epoch, idepth_scale)), step, epoch, global_step)
            if max_steps is not None and global_step >= max_steps:
                return epoch, global_step
    return num_epochs - 1, global_step

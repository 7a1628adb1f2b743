"""torch.autograd.Functions over the engine's forward/backward operators on c4 activations.

Training path of SURVEY.md section 8 (row a-10; reference train.py:164-310 trains with stock autograd).
torch supplies the tape, the optimizer and the elementwise glue (concatenation along channel groups,
sigmoid of the 1-channel heads, losses); every convolution, BatchNorm and resampling -- forward and
backward -- is a HIP kernel.  The plane sweep is a leaf without gradient.
"""
import os

import torch

from . import _lib, ops


def _s():
    return torch.cuda.current_stream().cuda_stream


WINOGRAD = True     # stride-1 3x3 / 5x5 / 7x7 convolutions (forward and data gradient) through the Winograd kernels


def _winograd_ok(k, stride, cout):
    return WINOGRAD and cout % 64 == 0 and ((stride == 1 and k == 3) or (stride in (1, 2) and k in (5, 7)))


_SYNC = {}      # (device, stream) -> sync workspace of the LDS-staged F(4x4,3x3) kernel: launches on one stream are ordered, so one per stream is enough


def _sync_workspace(device):
    key = (str(device), torch.cuda.current_stream(device).cuda_stream)
    ws = _SYNC.get(key)
    if ws is None:                                                       # inside a graph capture this allocates from the graph's pool; the zero fill is replayed with it
        ws = _SYNC[key] = ops.wino36_sync_workspace(device)
    return ws


_BN_WS = {}     # (device, stream) -> fp64 sum workspace of the BatchNorm kernels: zero between calls (the kernels clear what they read)
_BN_WS_DOUBLES = 8 * 512 * 4      # 8 per channel group and statistics group: up to 2048 channels x 4 groups


def _bn_workspace(device, G, groups=1):
    if 8 * G * groups > _BN_WS_DOUBLES:
        raise _lib.EngineError("BatchNorm over %d channel groups: workspace too small" % G)
    key = (str(device), torch.cuda.current_stream(device).cuda_stream)
    ws = _BN_WS.get(key)
    if ws is None:
        ws = _BN_WS[key] = torch.zeros(_BN_WS_DOUBLES, device=device, dtype=torch.float64)
    return ws


_BN_PART = {}   # (device, stream) -> fp64 partial-sum workspace of the two-launch BatchNorm (cnm_bn_train_*_p_c4_f32): contents irrelevant between calls
BN_PARTIALS = os.environ.get("CNM_BN_PARTIALS", "1") != "0"   # [r6] BatchNorm forward / backward in two launches each (fixed-slot partial sums) instead of three


def _bn_partials(device, N, C, H, W, groups):
    n = _lib.load().cnm_bn_train_partials_doubles(N, C, H, W, groups)
    key = (str(device), torch.cuda.current_stream(device).cuda_stream)
    ws = _BN_PART.get(key)
    if ws is None or ws.numel() < n:
        ws = _BN_PART[key] = torch.empty(max(n, 1 << 20), device=device, dtype=torch.float64)
    return ws


PACK_CACHE = None   # a dict while a trainer step runs (trainer._step_scope): packed filters shared by the two depthNet passes of a step
PACK_PLAN = None    # the PackPlan of the running step (trainer._step_scope), or None


class PackPlan:
    """[r6] The 36-point 3x3 filter packs of a training step as ONE launch.  Every filter is re-packed every step (the weights change in
    optimizer.step()), forward form and data-gradient form: 63 launches of ~10 us.  The first step under a plan RECORDS which packs it
    asked for ("u4": pack_winograd4(weight, None, rot), "u4d": pack_winograd4_dgrad(weight)); every later step runs them all with one
    cnm_pack_winograd4_batch_f32 launch when the step scope opens and hands the results out through PACK_CACHE.  A pack the plan does not
    know (another shape, another variant of the step) is computed the old way.  Outputs are persistent tensors: static under graph capture."""
    BATCH = os.environ.get("CNM_PACK_BATCH", "1") != "0"

    def __init__(self):
        self.requests, self.seen, self.table, self.outs, self.blocks = [], set(), None, None, 0

    def record(self, kind, weight, rot):
        key = (kind, id(weight), rot)
        if self.table is None and key not in self.seen and self.BATCH and kind in ("u4", "u4d") and weight.shape[2] == 3 and weight.is_cuda:
            self.seen.add(key)
            self.requests.append((kind, weight, rot))

    def freeze(self):
        """End of the recording step: allocate the outputs, build the job table on the device."""
        if self.table is not None or not self.requests:
            return
        import numpy as np
        lib = _lib.load()
        jobs = np.zeros(len(self.requests), dtype=np.dtype([("w", "<u8"), ("out", "<u8"), ("Cout", "<i4"), ("Cin", "<i4"), ("rot", "<i4"),
                                                            ("nchunks", "<i4"), ("dgrad", "<i4"), ("first", "<i4")]))
        assert jobs.dtype.itemsize == 40
        self.outs, first = [], 0
        for j, (kind, w, rot) in enumerate(self.requests):
            co, ci = (w.shape[1], w.shape[0]) if kind == "u4d" else (w.shape[0], w.shape[1])     # as the single-filter call sees them
            out = torch.empty(lib.cnm_packed_winograd4_floats(co, ci), device=w.device, dtype=torch.float32)
            nch = (4 * ((ci + 3) // 4) + 15) // 16
            jobs[j] = (w.data_ptr(), out.data_ptr(), co, ci, rot, nch, int(kind == "u4d"), first)
            first += nch * (co // 16)
            self.outs.append(out)
        self.blocks = first
        self.table = torch.from_numpy(jobs.view(np.uint8).copy()).to(self.requests[0][1].device)

    def run(self, launch=True):
        """Step scope opens: one launch for every recorded pack (launch=False: an earlier scope of this step has launched it), results
        into PACK_CACHE under the keys _packed() will ask for."""
        if self.table is None:
            return
        w0 = self.requests[0][1]
        if any((not w.is_contiguous()) or w.data_ptr() != int(p) for (_, w, _), p in zip(self.requests, self._ptrs())):
            self.__init__()                                               # a parameter moved (load_state_dict keeps storage; .to(), re-creation do not): record again
            return
        if launch:
            with torch.cuda.device(w0.device):
                _lib.check(_lib.load().cnm_pack_winograd4_batch_f32(self.table.data_ptr(), len(self.requests), self.blocks, _s()))
        for (kind, w, rot), out in zip(self.requests, self.outs):
            PACK_CACHE[(kind, w.data_ptr(), w._version, tuple(w.shape), 0 if kind == "u4d" else rot, 1)] = (w, out)

    def _ptrs(self):
        if not hasattr(self, "_ptr_cache") or self._ptr_cache[0] is not self.table:
            import numpy as np
            raw = self.table.cpu().numpy().view(np.dtype([("w", "<u8"), ("rest", "V32")]))
            self._ptr_cache = (self.table, [int(v) for v in raw["w"]])
        return self._ptr_cache[1]


def _packed(kind, weight, rot, stride, fn):
    """Packed form of `weight` for one step: the trainer opens a cache per step (forward AND backward: the weights only
    change in optimizer.step()), so depthNet's two passes (train.py:164-167) pack every filter once per direction instead
    of twice.  Outside a step scope nothing is cached (a HIP-graph replay updates weights without bumping _version)."""
    c = PACK_CACHE
    if c is None:
        return fn()
    if PACK_PLAN is not None:
        PACK_PLAN.record(kind, weight, rot)
    key = (kind, weight.data_ptr(), weight._version, tuple(weight.shape), rot, stride)
    hit = c.get(key)
    # the keyed tensor is held next to the value: a temporary used as a weight (a detached view, a derived filter) may be
    # freed during backward and a later temporary of the same shape can land on its address with version 0 --
    # while the cache holds the storage that cannot happen, and a hit is verified against the holder's storage anyway
    if hit is not None and hit[0].untyped_storage().data_ptr() == weight.untyped_storage().data_ptr():
        return hit[1]
    v = fn()
    c[key] = (weight, v)
    return v


def _dgrad_weight(weight):
    """w'[ci][co] = w[co][ci] rotated by 180 degrees: the data gradient of a stride-1 convolution is a convolution with w'."""
    return _packed("flip", weight, 0, 1, lambda: weight.detach().flip(2, 3).transpose(0, 1).contiguous())


def _winograd_conv(x, weight, rot, stride=1, dgrad=False):
    """conv(x, weight); dgrad: conv(x, w') with w'[ci][co] = w[co][ci] rotated by 180 degrees (the data gradient of a stride-1
    convolution with `weight`): the 36-point pack reads w' straight from `weight`, the other kernels get the flipped tensor."""
    Cout, _, k, _ = weight.shape
    if dgrad:
        Cout = weight.shape[1]
    # large layers: F(4x4,3x3); with the sync workspace the staged kernel also balances small unit counts (>= 3 x 3 tiles per image).
    # 5x5 stride 1 (conv2.0) with enough 2 x 2 tiles: F(2x2,5x5) on the same kernel -- relative L2 error against the fp64
    # convolution 2.0-2.6e-6, as F(4x4,3x3) (tools/rel_err_probe.py; rows F(2,5): 1.6-2.2e-6)
    if stride == 1 and ((k == 3 and (_winograd4_fills_chip(x, Cout) or (WINOGRAD4_SMALL and Cout % 128 == 0 and x.shape[2] >= 9 and x.shape[3] >= 9)))
                        or (k == 5 and FAST_FORWARD and _winograd4_fills_chip(x, Cout, 2))):
        if dgrad:
            up4 = _packed("u4d", weight, 0, 1, lambda: ops.pack_winograd4_dgrad(weight))
        else:
            up4 = _packed("u4", weight, rot, 1, lambda: ops.pack_winograd4(weight, None, rot))
        return ops.conv3x3_winograd4_c4(x, up4, None, Cout, relu=False, ksize=k, sync=_sync_workspace(x.device))
    if not dgrad and stride == 2 and _s2_phases_ok(x, Cout, k) and not (k == 3 and _rows3_stride2_ok(x, Cout)):
        # stride-2 5x5 / 7x7 forward (and the 3x3 ones the row kernel does not take): the four pixel phases of x on the staged
        # 36-point kernel (F(4x4,3x3) / F(3x3,4x4)), the transforms the 3x3 layers of the step already run with
        ups = _packed("u4s2", weight, rot, 2, lambda: ops.pack_winograd4_s2(weight, None, rot))
        return ops.conv_s2_winograd4_c4(x, ups, None, Cout, k, relu=False, sync=_sync_workspace(x.device))
    if k == 3 and stride == 2:                                           # (only called where _rows3_stride2_ok says so) two F(4,2) column phases along rows
        up = _packed("u4r", weight, rot, 2, lambda: ops.pack_winograd_rows(weight, None, rot, stride=2, tile=4))
        return ops.conv_rows_winograd_c4(x, up, None, Cout, 3, relu=False, stride=2, tile=4)
    if dgrad:
        weight = _dgrad_weight(weight)
    if k == 7 and stride == 1 and FAST_ROWS7 and Cout % 128 == 0:
        # conv1.0: F(4,7) along rows on the staged kernel (0.88 -> 0.54 ms for 4 pairs at 192x256, tools/rel_err_probe.py).  OFF:
        # its relative L2 error is 2.5-4.5e-6 on non-negative inputs (a cost volume) but 2.1e-5 on zero-mean random data, at the
        # op-level bar of tests/test_gpu_training.py::test_conv_forward_dgrad_wgrad (2e-5); F(2,7) stays a decade below it
        up = _packed("u4r", weight, rot, 1, lambda: ops.pack_winograd(weight, None, rot, stride=1, tile=4))
        return ops.conv_rows_winograd_c4(x, up, None, Cout, 7, relu=False, stride=1, tile=4, sync=_sync_workspace(x.device))
    up = _packed("u2", weight, rot, stride, lambda: ops.pack_winograd(weight, None, rot, stride=stride, tile=2))    # F(2,k) rows
    if k == 3:
        return ops.conv3x3_winograd_c4(x, up, None, Cout, relu=False)
    return ops.conv_rows_winograd_c4(x, up, None, Cout, k, relu=False, stride=stride, tile=2)


BN_RECOMPUTE_MASK = os.environ.get("CNM_BN_RECOMPUTE_MASK", "1") != "0"   # BatchNorm backward without the saved output (mask from x): 5 tensor passes instead of 7
FAST_FORWARD = True              # the inference kernels' larger tiles where their relative L2 error stays a decade below the op-level bar (2e-5): F(2x2,5x5), F(4,2) stride-2 rows
WINOGRAD_WGRAD = True            # weight gradient of the 3x3 stride-1 layers in the Winograd domain (transform, 36 GEMMs, inverse transform)
WINOGRAD_WGRAD_MIN_PIXELS = 512     # ... from this many pixels N*H*W on (4 x 12 x 16 = 768: 1.2-1.4x; 4 x 6 x 8: the direct kernel)
PAD_DGRAD = True                 # 3x3 stride-1 data gradients with ragged input-channel counts on the Winograd kernels (padded), not the direct kernel
FAST_ROWS7 = False               # ... F(4,7) for conv1.0 does not (see _winograd_conv): off
WINOGRAD4_MIN_WORKGROUPS = 384   # same switch point as the inference executors (include/cnm_engine.h)
S2_PHASE_KSIZES = (3, 5, 7)         # stride-2 layers whose FORWARD runs on the pixel phases of the input (ops.conv_s2_winograd4_c4); () keeps the F(2,k) row phases
S2_DGRAD_SCATTER = True          # stride-2 data gradient with 3x3 phase filters: one phase-interleaving launch instead of four convolutions + four strided copies
WINOGRAD4_SMALL = True           # ... and the same extension below it (nets.hip wino4_staged_small)


def _winograd4_fills_chip(x, Cout, m=4):
    N, _, H, W, _ = x.shape
    return (Cout // 64) * -(-(N * -(-H // m) * -(-W // m)) // 16) >= WINOGRAD4_MIN_WORKGROUPS


def _s2_phases_ok(x, Cout, k):
    return (WINOGRAD and k in S2_PHASE_KSIZES and x.shape[2] % 2 == 0 and x.shape[3] % 2 == 0
            and bool(_lib.load().cnm_conv_s2_winograd4_ok(Cout, x.shape[2], x.shape[3], k)))


def _rows3_stride2_ok(x, Cout):
    """3x3 stride 2 along rows (two F(4,2) column phases, 7.5 multiplies per output instead of 9): where the inference executor
    takes it (nets.hip EngF32::conv: at least 384 workgroups of 64 couts x 48 row tiles)."""
    N, _, H, W, _ = x.shape
    return FAST_FORWARD and WINOGRAD and Cout % 64 == 0 and (Cout // 64) * -(-(N * -(-H // 2) * -(-(-(-W // 2)) // 4)) // 48) >= 384


def _stride2_dgrad_phases(weight):
    """Data gradient of a stride-2 convolution (odd k, pad k//2, even H and W) as four stride-1 convolutions of dY,
    one per pixel phase (a, b) of dX -- no structural zeros:
        dX[2i+a, 2j+b] = sum_{u,v} dY[i + c_a - u, j + c_b - v] . w[:, :, r_a + 2u, r_b + 2v],   r = (phase + pad) % 2,
    c = (phase + pad - r) / 2.  Returns [(a, b, w')] with w' [Cin, Cout, K', K'] (K' = 3 or 5, zero-embedded) such that
    dX_phase = conv2d(dY, w', stride 1, pad K'//2)."""
    Cout, Cin, k, _ = weight.shape
    pad = k // 2
    taps = []
    for ph in (0, 1):
        r = (ph + pad) % 2
        c = (ph + pad - r) // 2
        U = len(range(r, k, 2))
        taps.append((r, c - (U - 1), c))                                 # sub-kernel rows r::2, tap u at offset c - u: offsets [c-U+1, c]
    wt = weight.transpose(0, 1)                                          # [Cin, Cout, k, k] view
    out = []
    for a in (0, 1):
        for b in (0, 1):
            (ra, loa, hia), (rb, lob, hib) = taps[a], taps[b]
            K = 3 if max(abs(loa), abs(hia), abs(lob), abs(hib)) <= 1 else 5
            wp = weight.new_zeros(Cin, Cout, K, K)
            # offsets descend with the tap index: the flipped sub-kernel fills one contiguous window of w'
            wp[:, :, loa + K // 2:hia + K // 2 + 1, lob + K // 2:hib + K // 2 + 1] = wt[:, :, ra::2, rb::2].flip(2, 3)
            out.append((a, b, wp))
    return out


_S2_CAT_INDEX = {}


def _stride2_dgrad_cat(weight):
    """The four phase filters of _stride2_dgrad_phases concatenated along the output channels, [4 * Cin, Cout, T, T] with taps at
    offsets -1 .. 1 (k = 3, 5: T = 3) or -1 .. 2 (k = 7: T = 4) -- what the phase-interleaving data-gradient launch packs -- built by
    ONE gather from the weight (index table cached per k) instead of a zero fill, a flip and a strided copy per phase."""
    Cout, Cin, k, _ = weight.shape
    T = 4 if k == 7 else 3
    key = (k, str(weight.device))
    idx = _S2_CAT_INDEX.get(key)
    if idx is None:
        pad, rows = k // 2, []
        taps = []
        for ph in (0, 1):
            r = (ph + pad) % 2
            taps.append((r, (ph + pad - r) // 2, len(range(r, k, 2))))    # sub-kernel rows r::2, tap u at offset c - u
        for a in (0, 1):
            for b in (0, 1):
                (ra, ca, Ua), (rb, cb, Ub) = taps[a], taps[b]
                for oy in range(-1, T - 1):
                    for ox in range(-1, T - 1):
                        u, v = ca - oy, cb - ox
                        rows.append((ra + 2 * u) * k + (rb + 2 * v) if 0 <= u < Ua and 0 <= v < Ub else k * k)   # k * k: the zero slot
        idx = _S2_CAT_INDEX[key] = torch.tensor(rows, device=weight.device, dtype=torch.long)
    wt = torch.nn.functional.pad(weight.transpose(0, 1).reshape(Cin, Cout, k * k), (0, 1))        # [Cin, Cout, k*k + 1], last slot zero
    return wt.index_select(2, idx).reshape(Cin, Cout, 4, T, T).permute(2, 0, 1, 3, 4).reshape(4 * Cin, Cout, T, T)


def _taps4(wp):
    """A phase filter of _stride2_dgrad_phases (3x3: offsets -1 .. 1, 5x5: offsets -2 .. 2 with nothing at -2) as 4x4 taps at
    offsets -1 .. 2."""
    if wp.shape[2] == 3:
        return torch.nn.functional.pad(wp, (0, 1, 0, 1))
    return wp[:, :, 1:, 1:]


def _wino_wgrad_fits(k, stride, N, H, W, Cin, Cout):
    """The Winograd-domain weight gradients address their transformed tensors with 32-bit byte offsets and count at most 2^24
    tiles (the entry points refuse larger problems): then the direct kernel runs."""
    gin, gout = -(-Cin // 4), Cout // 4
    if stride == 1 and k == 3:
        pts, t, geff = 36, N * -(-H // 4) * -(-W // 4), gin
    elif stride == 1:
        pts, t, geff = k + 3, N * H * -(-W // 4), gin
    else:
        m = 3 if k == 7 else 4
        pts, t, geff = 36, N * -(-(H // 2) // m) * -(-(W // 2) // m), 4 * gin
    return t < (1 << 24) and pts * t * max(geff, gout) * 16 < 0xFFFFFFFF


class ConvC4(torch.autograd.Function):
    """y = conv2d(x, weight, stride, padding=(k-1)//2) on c4 tensors, no bias.
    x [N,ceil(Cin/4),H,W,4] (channels possibly rotated by `rot`), weight OIHW; Cout % 4 == 0, >= 16."""

    @staticmethod
    def forward(ctx, x, weight, stride, rot):
        x = x.contiguous()
        Cout, Cin, k, _ = weight.shape
        if _winograd_ok(k, stride, Cout) or (k == 3 and stride == 2 and (_rows3_stride2_ok(x, Cout) or _s2_phases_ok(x, Cout, 3))):
            y = _winograd_conv(x, weight.detach(), rot, stride)
        elif (WINOGRAD and k == 3 and stride == 2 and Cout % 64 == 0
              and (Cout // 64) * -(-(x.shape[0] * -(-x.shape[2] // 2) * -(-x.shape[3] // 2)) // 64) < 256):
            # the deepest stride-2 layer (conv5.3: 48 implicit-GEMM tiles with a 4608-deep reduction each): F(2x2,3x3) keeping one output
            # per tile, as the inference executor does (nets.hip EngF32::conv)
            up = _packed("u2s2", weight, rot, 2, lambda: ops.pack_winograd(weight.detach(), None, rot, stride=1, tile=2))
            y = ops.conv3x3_s2_winograd_c4(x, up, None, Cout, relu=False)
        else:
            wp, _ = ops.pack_conv(weight.detach(), None, None, rot)
            y = ops.conv2d_c4(x, wp, None, Cout, k, stride, relu=False)
        ctx.save_for_backward(x, weight)
        ctx.stride, ctx.rot = stride, rot
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        dy = dy.contiguous()
        Cout, Cin, k, _ = weight.shape
        N, G, H, W, _ = x.shape
        lib, dev = _lib.load(), x.device
        dx = dw = None
        with torch.cuda.device(dev):
            rot = ctx.rot
            # input channels stored rotated (refine conv1.0: 64 features, then the 3 maps): x position j holds weight channel (j + rot) % Cin,
            # so the data gradient in x's order is the one of the weight with its input channels rolled by rot
            wdg = weight.detach() if rot == 0 or ctx.stride != 1 else _packed("roll", weight, rot, 1, lambda: torch.cat((weight.detach()[:, rot:], weight.detach()[:, :rot]), 1))
            if ctx.needs_input_grad[0] and ctx.stride == 1 and _winograd_ok(k, 1, Cin):
                # stride 1: dx = conv(dy, w') with w'[ci][co] = w[co][ci] rotated by 180 degrees -- the same Winograd kernels
                dx = _winograd_conv(dy, wdg, 0, dgrad=True)
            elif (ctx.needs_input_grad[0] and ctx.stride == 1 and k == 3 and PAD_DGRAD and _winograd_ok(3, 1, 64) and Cin > 64):
                # stride 1, input channels not a multiple of 64 (the concatenations with a disparity channel: 65, 257, 513): the same
                # F(4x4,3x3) data gradient with w' zero-padded to the next 64 output channels -- 1.1-2x the multiplies of a kernel
                # that needs a quarter of the direct count -- and the padding groups dropped (a view)
                Cp = 128 if Cin < 128 else 64 * -(-Cin // 64)
                wpad = _packed("wpad", weight, rot, 1, lambda: torch.nn.functional.pad(wdg, (0, 0, 0, 0, 0, Cp - Cin)))
                dx = _winograd_conv(dy, wpad, 0, dgrad=True)[:, :G]
            elif (ctx.needs_input_grad[0] and ctx.rot == 0 and ctx.stride == 2 and WINOGRAD and Cin % 64 == 0
                  and H % 2 == 0 and W % 2 == 0):
                # stride 2: four stride-1 Winograd convolutions of dY, one per pixel phase of dX (sub-pixel
                # decomposition: the zero-upsampled dY with its 3/4 structural zeros never exists)
                if S2_DGRAD_SCATTER and k in (3, 5):
                    # 3x3 and 5x5 filters: all four phase filters are 3x3 -- ONE F(4x4,3x3) launch with 4*Cin output channels whose
                    # store path interleaves the phases (the kernel of the fused up_conv layers, zero padding): no scatter copies
                    up = _packed("s2cat", weight, 0, 2, lambda: ops.pack_winograd4(_stride2_dgrad_cat(weight.detach())))
                    dx = ops.conv3x3_phase_scatter_c4(dy, up, Cin, sync=_sync_workspace(dev))
                elif (S2_DGRAD_SCATTER and k == 7 and Cin % 32 == 0 and (dy.shape[3] + 2) // 3 >= 6 and (dy.shape[2] + 2) // 3 >= 2):
                    # 7x7: phase filters of 3 or 4 taps per axis at offsets -1 .. 1 / -1 .. 2 -- all four as 4x4 filters on F(3x3,4x4),
                    # again one phase-interleaving launch (staged kernel)
                    up = _packed("s2cat", weight, 0, 2, lambda: ops.pack_winograd36(_stride2_dgrad_cat(weight.detach())))
                    dx = ops.conv3x3_phase_scatter_c4(dy, up, Cin, sync=_sync_workspace(dev), ksize=4)
                else:
                    dx = torch.empty_like(x)
                    for a, b, wp in _packed("s2", weight, 0, 2, lambda: _stride2_dgrad_phases(weight.detach())):
                        dx[:, :, a::2, b::2] = _winograd_conv(dy, wp, 0)
            elif ctx.needs_input_grad[0]:
                wd = torch.empty(lib.cnm_packed_dgrad_floats(Cout, Cin, k), device=dev, dtype=torch.float32)
                _lib.check(lib.cnm_pack_conv_dgrad_f32(weight.detach().contiguous().data_ptr(), Cout, Cin, k, ctx.rot, wd.data_ptr(), _s()))
                dx = torch.empty_like(x)
                _lib.check(lib.cnm_conv2d_dgrad_c4_f32(dy.data_ptr(), dy.shape[1], 0, Cout, dx.data_ptr(), G, 0, Cin,
                                                       wd.data_ptr(), N, H, W, k, ctx.stride, _s()))
            wino = WINOGRAD_WGRAD and _wino_wgrad_fits(k, ctx.stride, N, H, W, Cin, Cout)
            if ctx.needs_input_grad[1] and wino and k == 3 and ctx.stride == 1 and N * H * W >= WINOGRAD_WGRAD_MIN_PIXELS:
                # 3x3 stride 1: the gradient in the Winograd domain of the forward's F(4x4,3x3) -- 36 GEMMs over the tiles, a quarter
                # of the direct kernel's multiplies (cnm_conv3x3_wgrad_winograd_c4_f32): 1.5-2.4x from 4 x 24 x 32 pixels up
                # (tools/wgrad_wino_probe.py)
                ws = torch.empty(lib.cnm_conv3x3_wgrad_winograd_workspace_floats(Cout, Cin, N, H, W), device=dev, dtype=torch.float32)
                dw = torch.empty_like(weight)
                _lib.check(lib.cnm_conv3x3_wgrad_winograd_c4_f32(x.data_ptr(), G, 0, Cin, dy.data_ptr(), dy.shape[1], 0, Cout,
                                                                 dw.data_ptr(), ws.data_ptr(), ws.numel(), N, H, W, ctx.rot, _s()))
            elif ctx.needs_input_grad[1] and wino and k in (5, 7) and ctx.stride == 1 and N * H * W >= 8 * WINOGRAD_WGRAD_MIN_PIXELS:
                # 7x7 / 5x5 stride 1 (conv1.0, conv2.0): row-wise, in the domain of F(4,7) / F(4,5) -- ten / eight weight gradients with
                # k x 1 taps on [N][H][W/4] images, 17.5 / 10 multiplies per pixel instead of 49 / 25 (cnm_conv7x7_wgrad_winograd_c4_f32,
                # cnm_conv5x5_wgrad_winograd_c4_f32): conv1.0 1.64 -> 0.75 ms
                wsf, fn = ((lib.cnm_conv7x7_wgrad_winograd_workspace_floats, lib.cnm_conv7x7_wgrad_winograd_c4_f32) if k == 7 else
                           (lib.cnm_conv5x5_wgrad_winograd_workspace_floats, lib.cnm_conv5x5_wgrad_winograd_c4_f32))
                ws = torch.empty(wsf(Cout, Cin, N, H, W), device=dev, dtype=torch.float32)
                dw = torch.empty_like(weight)
                _lib.check(fn(x.data_ptr(), G, 0, Cin, dy.data_ptr(), dy.shape[1], 0, Cout, dw.data_ptr(), ws.data_ptr(), ws.numel(), N, H, W, ctx.rot, _s()))
            elif (ctx.needs_input_grad[1] and wino and k in (5, 7) and ctx.stride == 2 and H % 2 == 0 and W % 2 == 0
                  and N * H * W >= 4 * WINOGRAD_WGRAD_MIN_PIXELS):
                # 5x5 / 7x7 stride 2: the same on the four pixel phases of x (cnm_conv_s2_wgrad_winograd_c4_f32)
                ws = torch.empty(lib.cnm_conv_s2_wgrad_winograd_workspace_floats(Cout, Cin, k, N, H, W), device=dev, dtype=torch.float32)
                dw = torch.empty_like(weight)
                _lib.check(lib.cnm_conv_s2_wgrad_winograd_c4_f32(x.data_ptr(), G, 0, Cin, dy.data_ptr(), dy.shape[1], 0, Cout,
                                                                 dw.data_ptr(), ws.data_ptr(), ws.numel(), N, H, W, k, ctx.rot, _s()))
            elif ctx.needs_input_grad[1]:
                Ho, Wo = dy.shape[2], dy.shape[3]
                ws = torch.empty(lib.cnm_conv2d_wgrad_workspace_floats(Cout, Cin, k, N, Ho, Wo), device=dev, dtype=torch.float32)
                dw = torch.empty_like(weight)
                _lib.check(lib.cnm_conv2d_wgrad_c4_f32(x.data_ptr(), G, 0, Cin, dy.data_ptr(), dy.shape[1], 0, Cout,
                                                       dw.data_ptr(), ws.data_ptr(), ws.numel(), N, H, W, k, ctx.stride, ctx.rot, _s()))
        return dx, dw, None, None


class BatchNormReLUC4(torch.autograd.Function):
    """nn.BatchNorm2d in train mode followed (optionally) by ReLU; running stats updated in place."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, momentum, eps, relu, num_batches_tracked=None, groups=1):
        """groups > 1: sample n takes the batch statistics of the samples n' = n (mod groups) -- `groups` separate calls in one."""
        x = x.contiguous()
        N, G, H, W, _ = x.shape
        C = gamma.numel()
        lib, dev = _lib.load(), x.device
        y = torch.empty_like(x)
        mean = torch.empty(groups * C, device=dev, dtype=torch.float32)
        invstd = torch.empty_like(mean)
        with torch.cuda.device(dev):
            fwd = lib.cnm_bn_train_forward_p_c4_f32 if BN_PARTIALS else lib.cnm_bn_train_forward_zg_c4_f32
            ws = _bn_partials(dev, N, C, H, W, groups) if BN_PARTIALS else _bn_workspace(dev, G, groups)
            _lib.check(fwd(
                x.data_ptr(), gamma.detach().contiguous().data_ptr(), beta.detach().contiguous().data_ptr(),
                running_mean.data_ptr() if running_mean is not None else 0,
                running_var.data_ptr() if running_var is not None else 0, float(momentum), float(eps), int(relu),
                y.data_ptr(), mean.data_ptr(), invstd.data_ptr(), ws.data_ptr(),
                num_batches_tracked.data_ptr() if num_batches_tracked is not None else 0, N, C, H, W, groups, _s()))
        # the backward recomputes the ReLU mask from x (cnm_bn_train_backward_zgb_c4_f32): y is not kept on the tape for BatchNorm's
        # sake, and the backward kernels read 5 tensors instead of 7
        if BN_RECOMPUTE_MASK:
            ctx.save_for_backward(x, beta, gamma, mean, invstd)
        else:
            ctx.save_for_backward(x, y, gamma, mean, invstd)
        ctx.relu, ctx.groups, ctx.recompute = relu, groups, BN_RECOMPUTE_MASK
        # The recomputed mask is only right for a forward whose outputs came from bn_affine (train_ops.hip: bn_apply_kernel -- the ONE
        # producer of this Function's y).  A future fused forward must either call bn_affine or clear this tag and keep y.
        ctx.forward_affine = "bn_affine"
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, gamma, mean, invstd = ctx.saved_tensors
        dy = dy.contiguous()
        N, G, H, W, _ = x.shape
        C = gamma.numel()
        lib, dev = _lib.load(), x.device
        dx = torch.empty_like(x)
        dgamma, dbeta = torch.empty_like(gamma), torch.empty_like(gamma)
        with torch.cuda.device(dev):
            if ctx.recompute:
                assert getattr(ctx, "forward_affine", None) == "bn_affine", "the mask-recomputing backward needs the forward that rounds like bn_affine"
            if BN_PARTIALS:                                 # `y` holds beta when the mask is recomputed (see forward)
                _lib.check(lib.cnm_bn_train_backward_p_c4_f32(
                    x.data_ptr(), 0 if ctx.recompute else y.data_ptr(), dy.data_ptr(), gamma.detach().contiguous().data_ptr(),
                    y.detach().contiguous().data_ptr() if ctx.recompute else 0, mean.data_ptr(), invstd.data_ptr(),
                    int(ctx.relu), dx.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(), _bn_partials(dev, N, C, H, W, ctx.groups).data_ptr(), N, C, H, W, ctx.groups, _s()))
            elif ctx.recompute:
                _lib.check(lib.cnm_bn_train_backward_zgb_c4_f32(
                    x.data_ptr(), dy.data_ptr(), gamma.detach().contiguous().data_ptr(), y.detach().contiguous().data_ptr(), mean.data_ptr(), invstd.data_ptr(),
                    int(ctx.relu), dx.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(), _bn_workspace(dev, G, ctx.groups).data_ptr(), N, C, H, W, ctx.groups, _s()))
            else:
                _lib.check(lib.cnm_bn_train_backward_zg_c4_f32(
                    x.data_ptr(), y.data_ptr(), dy.data_ptr(), gamma.detach().contiguous().data_ptr(), mean.data_ptr(), invstd.data_ptr(),
                    int(ctx.relu), dx.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(), _bn_workspace(dev, G, ctx.groups).data_ptr(), N, C, H, W, ctx.groups, _s()))
        return dx, dgamma, dbeta, None, None, None, None, None, None, None


class Upsample2xC4(torch.autograd.Function):
    """nn.Upsample(scale_factor=2, mode='bilinear', align_corners=False) on c4 tensors."""

    @staticmethod
    def forward(ctx, x):
        return ops.upsample2x_c4(x.contiguous())

    @staticmethod
    def backward(ctx, dy):
        dy = dy.contiguous()
        N, G, Ho, Wo, _ = dy.shape
        dx = torch.empty(N, G, Ho // 2, Wo // 2, 4, device=dy.device, dtype=torch.float32)
        with torch.cuda.device(dy.device):
            _lib.check(_lib.load().cnm_upsample2x_backward_c4_f32(dy.data_ptr(), dx.data_ptr(), N, G, Ho // 2, Wo // 2, _s()))
        return dx


_ML1_WS = {}    # (device, stream) -> ticket + block partials of cnm_masked_l1_f32 (the ticket is zero between calls)


def _ml1_workspace(device):
    key = (str(device), torch.cuda.current_stream(device).cuda_stream)
    ws = _ML1_WS.get(key)
    if ws is None:
        ws = _ML1_WS[key] = torch.zeros(_lib.load().cnm_masked_l1_workspace_doubles(), device=device, dtype=torch.float64)
    return ws


class MaskedL1(torch.autograd.Function):
    """Masked mean L1 of the training losses (reference losses.py:30-73) as one launch each way: value =
    sum_m weight |pred - gt| / count(m) with m = gt > 0 & finite(gt) & finite(pred) & pred > 0; gradients for pred and weight.
    The torch expression of the same thing (trainer._masked_l1) is ~20 elementwise / reduction launches per term."""

    @staticmethod
    def forward(ctx, pred, gt, weight):
        pred, gt = pred.contiguous(), gt.contiguous()
        weight = weight.contiguous() if weight is not None else None
        assert pred.shape == gt.shape and (weight is None or weight.shape == pred.shape)
        out = torch.empty(2, device=pred.device, dtype=torch.float32)
        with torch.cuda.device(pred.device):
            _lib.check(_lib.load().cnm_masked_l1_f32(ops._p(pred), ops._p(gt), ops._p(weight), pred.numel(), ops._p(_ml1_workspace(pred.device)), ops._p(out), _s()))
        ctx.save_for_backward(pred, gt, weight, out)
        return out[0]

    @staticmethod
    def backward(ctx, go):
        pred, gt, weight, out = ctx.saved_tensors
        need_p, need_w = ctx.needs_input_grad[0], ctx.needs_input_grad[2] and weight is not None
        if not (need_p or need_w):
            return None, None, None
        dp = torch.empty_like(pred) if need_p else None
        dw = torch.empty_like(weight) if need_w else None
        go = go.contiguous().float()
        with torch.cuda.device(pred.device):
            _lib.check(_lib.load().cnm_masked_l1_backward_f32(ops._p(pred), ops._p(gt), ops._p(weight), ops._p(go), ops._p(out), pred.numel(), ops._p(dp), ops._p(dw), _s()))
        return dp, None, dw


class MaskedL1Both(torch.autograd.Function):
    """[r6] sum_m |a - b| / max(count(m), 1) with m = a > 0 & finite(a) & b > 0 & finite(b), gradients to BOTH arguments (the warped-depth loss of
    trainer.get_warped_depth_loss: the refined depth reaches it directly and through the sampling position of the warp).  MaskedL1's two
    kernels: its mask is symmetric, d / d b = -d / d a, and an empty mask gives 0 with zero gradients instead of the kernel's NaN."""

    @staticmethod
    def forward(ctx, a, b):
        a, b = a.contiguous(), b.contiguous()
        assert a.shape == b.shape
        out = torch.empty(2, device=a.device, dtype=torch.float32)
        with torch.cuda.device(a.device):
            _lib.check(_lib.load().cnm_masked_l1_f32(ops._p(a), ops._p(b), None, a.numel(), ops._p(_ml1_workspace(a.device)), ops._p(out), _s()))
        ctx.save_for_backward(a, b, out)
        return torch.where(out[1] > 0, out[0], torch.zeros((), device=a.device, dtype=torch.float32))

    @staticmethod
    def backward(ctx, go):
        a, b, out = ctx.saved_tensors
        da = torch.empty_like(a)
        go = go.contiguous().float()
        with torch.cuda.device(a.device):                                # an empty mask: every element is outside it and gets an exact zero
            _lib.check(_lib.load().cnm_masked_l1_backward_f32(ops._p(a), ops._p(b), None, ops._p(go), ops._p(out), a.numel(), ops._p(da), None, _s()))
        return (da if ctx.needs_input_grad[0] else None), (-da if ctx.needs_input_grad[1] else None)


class NormalCosTerms(torch.autograd.Function):
    """[r6] Per-sample terms of the surface-normal loss (reference losses.py:76-122 as train.py:226-263 uses it): (sum over kept pixels of
    1 - cos(pred, gt), kept pixels) with kept = valid & finite(gt) & finite(pred) -- trainer.TrainStep._normal_terms as two launches forward and
    one backward instead of ~40 torch launches per term (three terms per step)."""

    @staticmethod
    def forward(ctx, pred, gt, valid):
        pred, gt = pred.contiguous(), gt.contiguous()
        B, C, H, W = pred.shape
        assert C == 3 and gt.shape == pred.shape and valid.numel() == B * H * W and valid.dtype == torch.bool
        valid = valid.contiguous()
        lib = _lib.load()
        ws = torch.empty(lib.cnm_normal_cos_workspace_doubles(B), device=pred.device, dtype=torch.float64)
        s, c = torch.empty(B, device=pred.device, dtype=torch.float32), torch.empty(B, device=pred.device, dtype=torch.float32)
        with torch.cuda.device(pred.device):
            _lib.check(lib.cnm_normal_cos_terms_f32(ops._p(pred), ops._p(gt), ops._p(valid), B, H * W, ops._p(ws), ops._p(s), ops._p(c), _s()))
        ctx.save_for_backward(pred, gt, valid)
        ctx.mark_non_differentiable(c)
        return s, c

    @staticmethod
    def backward(ctx, gs, _gc):
        pred, gt, valid = ctx.saved_tensors
        if not ctx.needs_input_grad[0]:
            return None, None, None
        B, _, H, W = pred.shape
        dp = torch.empty_like(pred)
        gs = gs.contiguous().float()
        with torch.cuda.device(pred.device):
            _lib.check(_lib.load().cnm_normal_cos_terms_backward_f32(ops._p(pred), ops._p(gt), ops._p(valid), ops._p(gs), B, H * W, ops._p(dp), _s()))
        return dp, None, None


class SplitSources(torch.autograd.Function):
    """x [B * S, ...] with sample n belonging to source n % S -> S contiguous tensors [B, ...] (depthNet.forward_sources).
    As strided slices x[s::S] every source would cost autograd a zero fill of the whole tensor, a strided copy and an addition;
    here the backward pass interleaves the S gradients in one copy."""

    @staticmethod
    def forward(ctx, x, S):
        ctx.S, ctx.shape = S, x.shape
        v = x.view(x.shape[0] // S, S, *x.shape[1:])
        return tuple(v[:, s].contiguous() for s in range(S))

    @staticmethod
    def backward(ctx, *grads):
        if all(g is None for g in grads):
            return None, None
        ref = next(g for g in grads if g is not None)
        grads = [g if g is not None else torch.zeros_like(ref) for g in grads]
        return torch.stack(grads, 1).reshape(ctx.shape), None


class C4ToNCHW(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, channels):
        ctx.groups = x.shape[1]
        return ops.c4_to_nchw(x.contiguous(), channels)

    @staticmethod
    def backward(ctx, dy):
        return ops.nchw_to_c4(dy.contiguous()), None


class NCHWToC4(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        ctx.channels = x.shape[1]
        return ops.nchw_to_c4(x.contiguous())

    @staticmethod
    def backward(ctx, dy):
        return ops.c4_to_nchw(dy.contiguous(), ctx.channels)


class Depth2NormalFn(torch.autograd.Function):
    """Depth2normal.forward (reference depth_util.py:149-203) with its gradient w.r.t. the depth map."""

    @staticmethod
    def forward(ctx, depth, intrinsic_inv, k_size, input_is_idepth):
        depth, intrinsic_inv = depth.contiguous(), intrinsic_inv.contiguous()
        normal, points = ops.depth2normal(depth, intrinsic_inv, k_size, input_is_idepth)
        ctx.save_for_backward(depth, intrinsic_inv)
        ctx.k, ctx.inv = k_size, input_is_idepth
        return normal, points

    @staticmethod
    def backward(ctx, gn, gp):
        depth, kinv = ctx.saved_tensors
        B, H, W = depth.shape
        gd = torch.empty_like(depth)
        ws = torch.empty(9 * B * H * W, device=depth.device, dtype=torch.float32)
        gn = gn.contiguous() if gn is not None else torch.zeros(B, 3, H, W, device=depth.device)
        gp_ptr = gp.contiguous() if gp is not None else None
        with torch.cuda.device(depth.device):
            _lib.check(_lib.load().cnm_depth2normal_backward_f32(
                depth.data_ptr(), kinv.data_ptr(), gn.data_ptr(), gp_ptr.data_ptr() if gp_ptr is not None else 0,
                gd.data_ptr(), ws.data_ptr(), B, H, W, ctx.k, int(ctx.inv), _s()))
        return gd, None, None, None


class InverseWarpFn(torch.autograd.Function):
    """inverse_warp (reference inverse_warp.py:81-118) with its gradient w.r.t. the target depth."""

    @staticmethod
    def forward(ctx, feat, depth, pose, K, K_inv):
        feat, depth, pose, K, K_inv = (t.contiguous() for t in (feat, depth, pose, K, K_inv))
        ctx.save_for_backward(feat, depth, pose, K, K_inv)
        return ops.inverse_warp(feat, depth, pose, K, K_inv)

    @staticmethod
    def backward(ctx, gout):
        feat, depth, pose, K, K_inv = ctx.saved_tensors
        B, C, H, W = feat.shape
        gd = torch.empty_like(depth)
        with torch.cuda.device(depth.device):
            _lib.check(_lib.load().cnm_inverse_warp_backward_depth_f32(
                feat.data_ptr(), depth.data_ptr(), pose.data_ptr(), K.data_ptr(), K_inv.data_ptr(),
                gout.contiguous().data_ptr(), gd.data_ptr(), B, C, H, W, _s()))
        return None, gd, None, None, None


# ---------------------------------------------------------------- building blocks used by the modules in train mode
def conv_bn_relu(x, conv, bn, rot=0, groups=1):
    """Conv2d(bias=False) -> BatchNorm2d(train) -> ReLU, as the reference's layer builders
    (depthNet_model.py:19-112) in training mode.  groups: statistics groups of the BatchNorm (sample n -> group n % groups)."""
    y = ConvC4.apply(x, conv.weight, conv.stride[0], rot)
    # num_batches_tracked (int64 on the device) is incremented by the forward kernel
    return BatchNormReLUC4.apply(y, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.momentum, bn.eps, True, bn.num_batches_tracked, groups)


FUSED_HEAD = True   # the one-channel heads through the head kernels (HeadC4) instead of an MFMA convolution padded to 16 output channels


class HeadC4(torch.autograd.Function):
    """depth_layer (reference depthNet_model.py:82-84, :246): scale * sigmoid(conv3x3(x; weight [1,C,3,3]) + bias) on a c4 tensor
    -> [N,1,H,W].  Forward is the inference head kernel (cnm_head_sigmoid_c4_f32); backward two streaming kernels
    (cnm_head_backward_c4_f32): as an MFMA problem padded to 16 output channels the layer moved 16x its data each way."""

    @staticmethod
    def forward(ctx, x, weight, bias, scale):
        x = x.contiguous()
        d = ops.head_sigmoid_c4(x, ops.pack_head(weight.detach()), bias.detach(), scale)
        ctx.save_for_backward(x, weight, d)
        ctx.scale = float(scale)
        return d

    @staticmethod
    def backward(ctx, gd):
        x, weight, d = ctx.saved_tensors
        N, G, H, W, _ = x.shape
        lib = _lib.load()
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        need_w = ctx.needs_input_grad[1] or ctx.needs_input_grad[2]
        dw = torch.empty_like(weight) if need_w else None
        db = torch.empty(1, device=x.device, dtype=torch.float32) if need_w else None
        ws = torch.empty(lib.cnm_head_backward_workspace_doubles(4 * G), device=x.device, dtype=torch.float64)
        with torch.cuda.device(x.device):
            _lib.check(lib.cnm_head_backward_c4_f32(ops._p(x), G, 0, 4 * G, ops._p(weight.detach().contiguous()), ops._p(gd.contiguous()), ops._p(d),
                                                    ctx.scale, ops._p(dx), ops._p(dw), ops._p(db), ops._p(ws), N, H, W, _s()))
        return dx, dw if ctx.needs_input_grad[1] else None, db if ctx.needs_input_grad[2] else None, None


def head(x, conv, scale):
    """depth_layer: Conv2d(C,1,3,padding=1) + Sigmoid, times `scale` (depthNet_model.py:82-84,246).
    FUSED_HEAD off: the 1-output-channel conv runs through the MFMA kernel padded to 16 output channels (zero rows)."""
    if FUSED_HEAD and x.is_cuda and x.shape[1] * 4 == conv.weight.shape[1]:
        return HeadC4.apply(x, conv.weight, conv.bias, float(scale))
    w = conv.weight
    w16 = torch.cat((w, w.new_zeros(15, *w.shape[1:])), 0)
    y = ConvC4.apply(x, w16, 1, 0)                        # [N,4,H,W,4]; channel 0 is the real one
    return scale * torch.sigmoid(y[:, 0, :, :, 0] + conv.bias).unsqueeze(1)      # [N,1,H,W]


def scalar_maps_to_group(*maps):
    """Up to four [N,1,H,W] maps -> one c4 channel group [N,1,H,W,4] (missing lanes zero)."""
    m = [t.squeeze(1) for t in maps]
    while len(m) < 4:
        m.append(torch.zeros_like(m[0]))
    return torch.stack(m, -1).unsqueeze(1)


def nearest_up2(d):
    """F.upsample(disp, scale_factor=2) (nearest; depthNet_model.py:247,252,257)."""
    return d.repeat_interleave(2, 2).repeat_interleave(2, 3)

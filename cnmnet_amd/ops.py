"""Operator-level Python API over the C ABI: torch tensors in, torch tensors out.

torch is used for device memory and streams only; every computation is a call into
libcnm_engine.so on the tensor's device and the current torch stream.  CPU tensors go to
the library's host twins (cnmnet_amd.host: the `_cpu` entry points, inference only) for the
operators that have one -- geometry prep, plane sweep, layout converters, depth->normal,
inverse warp -- and are rejected everywhere else; the CPU restatement in oracle/ is test
infrastructure and never involved.
"""
import weakref

import os
import torch

from . import _lib, host

IDEPTH_RANGE = {2.0: (0.02, 2.0), 3.0: (0.1, 3.0)}   # reference depthnet/depthNet_model.py:186-191


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _dev(*tensors):
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise _lib.EngineError("cnmnet_amd operators run on the GPU only; got a %s tensor" % t.device)
        if t.dtype != torch.float32:
            raise _lib.EngineError("cnmnet_amd operators take float32 tensors; got %s" % t.dtype)


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


def _p(t):
    return 0 if t is None else t.data_ptr()


def idepth_range(idepth_scale):
    try:
        return IDEPTH_RANGE[float(idepth_scale)]
    except KeyError:
        raise ValueError("idepth_scale must be 2.0 or 3.0 (got %r); the reference leaves the sweep range "
                         "undefined otherwise (depthNet_model.py:186-191)" % (idepth_scale,))


def homography_terms(ref_cam, src_cam):
    """ref_cam [B,2,4,4], src_cam [B,S,2,4,4] -> hmkt [B*S,12]   (depth_util.py:24-56)"""
    if host.is_host(ref_cam, src_cam):
        return host.homography_terms(ref_cam, src_cam)
    _dev(ref_cam, src_cam)
    ref_cam, src_cam = _c(ref_cam), _c(src_cam)
    B, S = src_cam.shape[0], src_cam.shape[1]
    out = torch.empty(B * S, 12, device=ref_cam.device, dtype=torch.float32)
    with torch.cuda.device(ref_cam.device):
        _lib.check(_lib.load().cnm_homography_terms_f32(_p(ref_cam), _p(src_cam), _p(out), B, S, _stream()))
    return out


def plane_sweep_volume(left, right, left_cam, right_cam, idepth_scale=3.0, planes=64):
    """Drop-in for depthNet.getVolume fed by process_camera_parameters
    (depthNet_model.py:185-224): left/right [B,3,H,W], cams [B,2,4,4] -> [B,planes,H,W]."""
    if host.is_host(left, right, left_cam, right_cam):
        return plane_sweep_volume_hmkt(left, right, homography_terms(left_cam, right_cam.unsqueeze(1)), idepth_scale, planes)
    _dev(left, right, left_cam, right_cam)
    return plane_sweep_volume_hmkt(left, right, homography_terms(left_cam, right_cam.unsqueeze(1)), idepth_scale, planes)


def plane_sweep_volume_hmkt(left, right, hmkt, idepth_scale=3.0, planes=64):
    """getVolume from the 12 camera terms per pair: left/right [B,3,H,W], hmkt [B,12] -> [B,planes,H,W]."""
    if host.is_host(left, right, hmkt):
        return host.plane_sweep_volume_hmkt(left, right, hmkt, *idepth_range(idepth_scale), planes)
    _dev(left, right, hmkt)
    lo, hi = idepth_range(idepth_scale)
    left, right, hmkt = _c(left), _c(right), _c(hmkt)
    B, _, H, W = left.shape
    vol = torch.empty(B, planes, H, W, device=left.device, dtype=torch.float32)
    lib = _lib.load()
    ws = torch.zeros(lib.cnm_planesweep_workspace_floats(B, 1, H, W), device=left.device, dtype=torch.float32)   # tile queue: zero on entry
    with torch.cuda.device(left.device):
        _lib.check(lib.cnm_planesweep_volume_nchw_f32(_p(left), _p(right), _p(hmkt), _p(vol), _p(ws), ws.numel(),
                                                      B, 1, H, W, planes, lo, hi, _stream()))
    return vol


def plane_sweep_cat_c4(ref, src, hmkt, idepth_scale=3.0, planes=64, ws=None, out=None):
    """ref [B,3,H,W], src [B,S,3,H,W], hmkt [B*S,12] -> c4 conv input [B*S, planes/4+1, H, W, 4]."""
    _dev(ref, src, hmkt, out)
    lo, hi = idepth_range(idepth_scale)
    ref, src = _c(ref), _c(src)
    B, S, _, H, W = src.shape
    x = out if out is not None else torch.empty(B * S, planes // 4 + 1, H, W, 4, device=ref.device, dtype=torch.float32)
    assert x.shape == (B * S, planes // 4 + 1, H, W, 4) and x.is_contiguous()
    lib = _lib.load()
    if ws is None:
        ws = torch.zeros(lib.cnm_planesweep_workspace_floats(B, S, H, W), device=ref.device, dtype=torch.float32)   # tile queue: zero on entry
    with torch.cuda.device(ref.device):
        _lib.check(lib.cnm_planesweep_cat_c4_f32(_p(ref), _p(src), _p(hmkt), _p(x), _p(ws), ws.numel(),
                                                 B, S, H, W, planes, lo, hi, _stream()))
    return x


def nchw_to_c4(x):
    if host.is_host(x):
        return host.nchw_to_c4(x)
    _dev(x)
    x = _c(x)
    N, Cc, H, W = x.shape
    G = (Cc + 3) // 4
    out = torch.empty(N, G, H, W, 4, device=x.device, dtype=torch.float32)
    with torch.cuda.device(x.device):
        _lib.check(_lib.load().cnm_nchw_to_c4_f32(_p(x), _p(out), G, 0, N, Cc, H, W, _stream()))
    return out


def c4_to_nchw(x, channels=None):
    if host.is_host(x):
        return host.c4_to_nchw(x, channels)
    _dev(x)
    N, G, H, W, _ = x.shape
    Cc = channels or 4 * G
    out = torch.empty(N, Cc, H, W, device=x.device, dtype=torch.float32)
    with torch.cuda.device(x.device):
        _lib.check(_lib.load().cnm_c4_to_nchw_f32(_p(_c(x)), G, 0, _p(out), N, Cc, H, W, _stream()))
    return out


def pack_conv(weight, bn=None, bias=None, rot=0, eps=1e-5):
    """weight [Cout,Cin,k,k]; bn = (gamma, beta, mean, var) or None -> (w_packed, b_packed)."""
    _dev(weight, bias, *(bn or ()))
    lib = _lib.load()
    Cout, Cin, k, _ = weight.shape
    wp = torch.empty(lib.cnm_packed_conv_floats(Cout, Cin, k), device=weight.device, dtype=torch.float32)
    if bias is not None:
        bias = _c(bias)
    bp = torch.empty(Cout, device=weight.device, dtype=torch.float32)
    g, b, m, v = [_c(t) for t in bn] if bn else (None, None, None, None)
    with torch.cuda.device(weight.device):
        _lib.check(lib.cnm_pack_conv_bn_f32(_p(_c(weight)), _p(g), _p(b), _p(m), _p(v), _p(bias), eps,
                                            Cout, Cin, k, rot, _p(wp), _p(bp), _stream()))
    return wp, bp


def pack_conv_f16(weight, bn=None, bias=None, rot=0, eps=1e-5):
    """fp16 twin of pack_conv: -> (w_packed half tensor, b_packed float tensor)."""
    _dev(weight, bias, *(bn or ()))
    lib = _lib.load()
    Cout, Cin, k, _ = weight.shape
    wp = torch.empty(lib.cnm_packed_conv_halfs(Cout, Cin, k), device=weight.device, dtype=torch.float16)
    bp = torch.empty(Cout, device=weight.device, dtype=torch.float32)
    g, b, m, v = [_c(t) for t in bn] if bn else (None, None, None, None)
    with torch.cuda.device(weight.device):
        _lib.check(lib.cnm_pack_conv_bn_f16(_p(_c(weight)), _p(g), _p(b), _p(m), _p(v), _p(bias), eps,
                                            Cout, Cin, k, rot, wp.data_ptr(), _p(bp), _stream()))
    return wp, bp


def nchw_to_c8(x):
    """fp32 NCHW -> fp16 c8 [N,ceil(C/8),H,W,8]."""
    _dev(x)
    x = _c(x)
    N, Cc, H, W = x.shape
    G = (Cc + 7) // 8
    out = torch.empty(N, G, H, W, 8, device=x.device, dtype=torch.float16)
    with torch.cuda.device(x.device):
        _lib.check(_lib.load().cnm_nchw_to_c8_f16(_p(x), out.data_ptr(), G, 0, N, Cc, H, W, _stream()))
    return out


def c8_to_nchw(x, channels=None):
    """fp16 c8 -> fp32 NCHW."""
    N, G, H, W, _ = x.shape
    Cc = channels or 8 * G
    out = torch.empty(N, Cc, H, W, device=x.device, dtype=torch.float32)
    with torch.cuda.device(x.device):
        _lib.check(_lib.load().cnm_c8_to_nchw_f16(x.contiguous().data_ptr(), G, 0, _p(out), N, Cc, H, W, _stream()))
    return out


def conv2d_c8(x, w_packed, b_packed, Cout, ksize, stride=1, relu=True):
    """fp16 c8 conv: x [N,G,H,W,8] half -> [N,Cout/8,Ho,Wo,8] half."""
    N, G, H, W, _ = x.shape
    pad = (ksize - 1) // 2
    Ho, Wo = (H + 2 * pad - ksize) // stride + 1, (W + 2 * pad - ksize) // stride + 1
    out = torch.empty(N, Cout // 8, Ho, Wo, 8, device=x.device, dtype=torch.float16)
    with torch.cuda.device(x.device):
        _lib.check(_lib.load().cnm_conv2d_c8_f16(x.data_ptr(), G, 0, G, out.data_ptr(), Cout // 8, 0, Cout, w_packed.data_ptr(), _p(b_packed),
                                                 N, H, W, ksize, stride, int(relu), _stream()))
    return out


def pack_head(weight):
    _dev(weight)
    Cc = weight.shape[1]
    wh = torch.empty(9 * Cc, device=weight.device, dtype=torch.float32)
    with torch.cuda.device(weight.device):
        _lib.check(_lib.load().cnm_pack_head_f32(_p(_c(weight)), Cc, _p(wh), _stream()))
    return wh


def conv2d_c4(x, w_packed, b_packed, Cout, ksize, stride=1, relu=True, x2=None):
    """x [N,G,H,W,4] (optionally concatenated with x2 along channels) -> [N,Cout/4,Ho,Wo,4]."""
    _dev(x, w_packed, b_packed, x2)
    N, G, H, W, _ = x.shape
    pad = (ksize - 1) // 2
    Ho, Wo = (H + 2 * pad - ksize) // stride + 1, (W + 2 * pad - ksize) // stride + 1
    out = torch.empty(N, Cout // 4, Ho, Wo, 4, device=x.device, dtype=torch.float32)
    lib = _lib.load()
    with torch.cuda.device(x.device):
        if x2 is None:
            _lib.check(lib.cnm_conv2d_c4_f32(_p(x), G, 0, G, _p(out), Cout // 4, 0, Cout, _p(w_packed), _p(b_packed),
                                             N, H, W, ksize, stride, int(relu), _stream()))
        else:
            G2 = x2.shape[1]
            _lib.check(lib.cnm_conv2d_cat2_c4_f32(_p(x), G, 0, G, _p(x2), G2, 0, G2, _p(out), Cout // 4, 0, Cout,
                                                  _p(w_packed), _p(b_packed), N, H, W, ksize, stride, int(relu), _stream()))
    return out


def rows_tile(ksize, stride, fast=True):
    """Outputs per tile of the row-wise Winograd kernels: 4 when `fast` (F(4,7); stride 2: F(4,4) / F(4,3) column phases)
    except for the 5-tap stride-1 rows (that layer runs on the 36-point 2-D kernel), else 2."""
    return 4 if (fast and not (ksize == 5 and stride == 1)) else 2


def pack_winograd(weight, bn=None, rot=0, eps=1e-5, stride=1, tile=None):
    """weight [Cout,Cin,k,k] (+ BatchNorm gamma/var fold) -> Winograd-domain packed filter:
    k = 3 (stride 1): F(2x2,3x3); k = 5, 7 (stride 1 or 2): row-wise F(2,k) / two F(2,ceil(k/2)) column phases."""
    _dev(weight, *(bn or ()))
    lib = _lib.load()
    Cout, Cin, k, _ = weight.shape
    assert k in (3, 5, 7) and stride in (1, 2)                          # k = 3: one filter serves stride 1 and the stride-2 mode
    tile = rows_tile(k, stride) if tile is None else tile
    n = lib.cnm_packed_winograd_floats(Cout, Cin) if k == 3 else lib.cnm_packed_winograd_rows_floats(Cout, Cin, k, stride, tile)
    up = torch.empty(n, device=weight.device, dtype=torch.float32)
    g, v = (_c(bn[0]), _c(bn[3])) if bn else (None, None)
    with torch.cuda.device(weight.device):
        if k == 3:
            _lib.check(lib.cnm_pack_winograd_bn_f32(_p(_c(weight)), _p(g), _p(v), eps, Cout, Cin, rot, _p(up), _stream()))
        else:
            _lib.check(lib.cnm_pack_winograd_rows_bn_f32(_p(_c(weight)), _p(g), _p(v), eps, Cout, Cin, k, stride, tile, rot, _p(up), _stream()))
    return up


def pack_winograd_rows(weight, bn=None, rot=0, eps=1e-5, stride=2, tile=4):
    """Row-wise Winograd filter for any supported (ksize, stride, tile) -- in particular 3x3 stride 2 (two F(4,2) column
    phases), which pack_winograd serves with the F(2x2,3x3) filter instead."""
    _dev(weight, *(bn or ()))
    lib = _lib.load()
    Cout, Cin, k, _ = weight.shape
    n = lib.cnm_packed_winograd_rows_floats(Cout, Cin, k, stride, tile)
    if n == 0:
        raise _lib.EngineError("no row-wise Winograd kernel for ksize %d stride %d tile %d (Cout %d)" % (k, stride, tile, Cout))
    up = torch.empty(n, device=weight.device, dtype=torch.float32)
    g, v = (_c(bn[0]), _c(bn[3])) if bn else (None, None)
    with torch.cuda.device(weight.device):
        _lib.check(lib.cnm_pack_winograd_rows_bn_f32(_p(_c(weight)), _p(g), _p(v), eps, Cout, Cin, k, stride, tile, rot, _p(up), _stream()))
    return up


def pack_winograd4(weight, bn=None, rot=0, eps=1e-5):
    """36-point Winograd-domain packed filter: 3x3 weight -> F(4x4,3x3); 5x5 weight -> F(2x2,5x5)."""
    _dev(weight, *(bn or ()))
    lib = _lib.load()
    Cout, Cin, k, _ = weight.shape
    assert k in (3, 5)
    up = torch.empty(lib.cnm_packed_winograd4_floats(Cout, Cin), device=weight.device, dtype=torch.float32)
    g, v = (_c(bn[0]), _c(bn[3])) if bn else (None, None)
    fn = lib.cnm_pack_winograd4_bn_f32 if k == 3 else lib.cnm_pack_winograd5x5_bn_f32
    with torch.cuda.device(weight.device):
        _lib.check(fn(_p(_c(weight)), _p(g), _p(v), eps, Cout, Cin, rot, _p(up), _stream()))
    return up


def pack_winograd4_s2(weight, bn=None, rot=0, eps=1e-5):
    """36-point packed filter of a STRIDE-2 3x3 / 5x5 / 7x7 layer as a stride-1 convolution of the four pixel phases of its input:
    5x5 (3x3) -> four 3x3 phase filters (F(4x4,3x3)), 7x7 -> four 4x4 phase filters (F(3x3,4x4))."""
    _dev(weight, *(bn or ()))
    lib = _lib.load()
    Cout, Cin, k, _ = weight.shape
    assert k in (3, 5, 7)
    up = torch.empty(lib.cnm_packed_winograd4_s2_floats(Cout, Cin), device=weight.device, dtype=torch.float32)
    g, v = (_c(bn[0]), _c(bn[3])) if bn else (None, None)
    with torch.cuda.device(weight.device):
        _lib.check(lib.cnm_pack_winograd4_s2_bn_f32(_p(_c(weight)), _p(g), _p(v), eps, Cout, Cin, k, rot, _p(up), _stream()))
    return up


def conv_s2_winograd4_c4(x, u_packed, b_packed, Cout, ksize, relu=True, x2=None, sync=None):
    """Stride-2 3x3 (pad 1) / 5x5 (pad 2) / 7x7 (pad 3) convolution on the LDS-staged 36-point kernel (pack_winograd4_s2 filter):
    x [N,G,H,W,4] (H, W even) -> [N,Cout/4,H/2,W/2,4].  sync = a wino36_sync_workspace (required)."""
    _dev(x, u_packed, b_packed, x2, sync)
    N, G, H, W, _ = x.shape
    out = torch.empty(N, Cout // 4, H // 2, W // 2, 4, device=x.device, dtype=torch.float32)
    G2 = x2.shape[1] if x2 is not None else 0
    with torch.cuda.device(x.device):
        _lib.check(_lib.load().cnm_conv_s2_winograd4_sync_c4_f32(_p(x), G, 0, G, _p(x2) if x2 is not None else None, G2, 0, G2,
                                                                 _p(out), Cout // 4, 0, Cout, _p(u_packed), _p(b_packed), N, H, W, ksize, int(relu),
                                                                 _p(sync) if sync is not None else None, sync.numel() if sync is not None else 0, _stream()))
    return out


def pack_winograd36(weight):
    """Plain 36-point pack of a [Cout,Cin,k,k] filter, k = 3 / 4 / 5 (F(4x4,3x3) / F(3x3,4x4) / F(2x2,5x5))."""
    _dev(weight)
    lib = _lib.load()
    Cout, Cin, k, _ = weight.shape
    up = torch.empty(lib.cnm_packed_winograd4_floats(Cout, Cin), device=weight.device, dtype=torch.float32)
    with torch.cuda.device(weight.device):
        _lib.check(lib.cnm_pack_winograd36_f32(_p(_c(weight)), Cout, Cin, k, _p(up), _stream()))
    return up


def conv3x3_phase_scatter_c4(x, u_packed, Cout, sync=None, out=None, ksize=3):
    """out[:, :, a::2, b::2] = conv(x, w_phase[2a + b]) (zero padding, no bias) for the four filters packed phase major as
    4*Cout output channels: x [N,G,H,W,4] -> [N,Cout/4,2H,2W,4], one launch.  ksize 3: 3x3 phase filters, taps -1 .. 1
    (pack_winograd4 of the [4*Cout, Cin, 3, 3] tensor); ksize 4: 4x4 phase filters, taps -1 .. 2 (pack_winograd36)."""
    _dev(x, u_packed, sync)
    N, G, H, W, _ = x.shape
    if out is None:
        out = torch.empty(N, Cout // 4, 2 * H, 2 * W, 4, device=x.device, dtype=torch.float32)
    fn = _lib.load().cnm_conv3x3_phase_scatter_winograd4_sync_c4_f32 if ksize == 3 else _lib.load().cnm_conv4x4_phase_scatter_winograd_sync_c4_f32
    with torch.cuda.device(x.device):
        _lib.check(fn(_p(x), G, 0, G, _p(out), Cout // 4, 0, Cout, _p(u_packed), None,
                                                                               N, H, W, 0, _p(sync) if sync is not None else None,
                                                                               sync.numel() if sync is not None else 0, _stream()))
    return out


def pack_winograd4_dgrad(weight):
    """pack_winograd4(weight.flip(2, 3).transpose(0, 1)) -- the 36-point filter of the data gradient of a stride-1 3x3 / 5x5
    convolution -- without materialising the flipped tensor."""
    _dev(weight)
    lib = _lib.load()
    Cout, Cin, k, _ = weight.shape
    assert k in (3, 5) and Cin % 64 == 0
    up = torch.empty(lib.cnm_packed_winograd4_floats(Cin, Cout), device=weight.device, dtype=torch.float32)
    with torch.cuda.device(weight.device):
        _lib.check(lib.cnm_pack_winograd4_dgrad_f32(_p(_c(weight)), Cout, Cin, k, _p(up), _stream()))
    return up


def conv3x3_s2_winograd_c4(x, u_packed, b_packed, Cout, relu=True):
    """3x3 stride-2 pad-1 conv through the F(2x2,3x3) kernel (element (0,0) of every tile): -> [N,Cout/4,ceil(H/2),ceil(W/2),4]."""
    _dev(x, u_packed, b_packed)
    N, G, H, W, _ = x.shape
    out = torch.empty(N, Cout // 4, (H + 1) // 2, (W + 1) // 2, 4, device=x.device, dtype=torch.float32)
    with torch.cuda.device(x.device):
        _lib.check(_lib.load().cnm_conv3x3_s2_winograd_c4_f32(_p(x), G, 0, G, _p(out), Cout // 4, 0, Cout, _p(u_packed), _p(b_packed),
                                                              N, H, W, int(relu), _stream()))
    return out


def wino36_sync_workspace(device):
    """Zeroed sync workspace for the LDS-staged persistent kernels (flag / generation words + partial-output slots,
    csrc/sync_ws.h): one per stream of launches; zero before the first use, reusable after every call."""
    ws = torch.zeros(_lib.load().cnm_wino36_sync_floats(), device=device, dtype=torch.float32)
    register_sync_owner(ws)
    return ws


_SYNC_OWNERS = weakref.WeakSet()    # tensors that hold sync words of the staged kernels (sync workspaces, the nets' workspaces)


def register_sync_owner(t):
    """Remember a tensor whose contents include stream-K flag words, so that engine_status(clear=True) can scrub it after a time-out."""
    _SYNC_OWNERS.add(t)
    return t


def sync_workspace_state(sync):
    """Number of flag words of a sync workspace that are not zero (csrc/sync_ws.h): between launches every flag has been re-armed
    by the range that consumed it -- 0 -- unless a hand-off timed out and its publisher came late."""
    return int((sync[:1024].view(torch.int32)[:1020] != 0).sum().item())


def engine_status(clear=True, device=None):
    """Raise EngineError if a stream-K hand-off has timed out since the last acknowledgement (cnm_engine_status; the status word is
    per device).  device=None asks every device that owns a registered workspace plus the current one (a net may live on a device that
    is not current: ADVICE r5), a device asks that one.  With clear=True the failure is acknowledged, so later launches are accepted
    again, and every registered workspace on that device is zeroed: a timed-out hand-off leaves stale generation words behind, which
    the queue that wrote them can never mistake for its own, but a workspace may next be used from another stream, i.e. another
    hardware queue whose dispatch counter runs through the same small numbers.  Synchronise first."""
    lib = _lib.load()
    if not torch.cuda.is_available():
        return _lib.check(lib.cnm_engine_status(int(bool(clear))))
    if device is None:
        devs = sorted({torch.cuda.current_device()} | {t.device.index for t in list(_SYNC_OWNERS) if t.is_cuda})
    else:
        devs = [torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device()]
    first = 0
    for d in devs:
        with torch.cuda.device(d):
            rc = lib.cnm_engine_status(int(bool(clear)))
            if rc != 0 and clear:
                torch.cuda.synchronize()
                for t in list(_SYNC_OWNERS):
                    if t.is_cuda and t.device.index == d:
                        t.zero_()
                torch.cuda.synchronize()
        first = first or rc
    _lib.check(first)


_SWEEP_STORE_CALIBRATED = set()


def calibrate_sweep_store(device=None, force=False):
    """The plane sweep's output-store policy for `device`, measured once per device and process (cnm_calibrate_sweep_store: 24 launches on
    ~630 MB of scratch, blocking) unless CNM_SWEEP_STORE or cnm_tune_sweep_store has already decided.  Never call under stream capture; the
    depthNet modules call it when they first allocate a workspace, i.e. before any graph is captured over them and before any timed region.
    Returns the policy in force (0 plain, 2 non-temporal)."""
    lib = _lib.load()
    dev = torch.device("cuda" if device is None else device)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    with torch.cuda.device(idx):
        pol = lib.cnm_tune_sweep_store(99, None)
        if (pol >= 0 or idx in _SWEEP_STORE_CALIBRATED) and not force:
            return pol
        if torch.cuda.is_current_stream_capturing():
            return pol
        _SWEEP_STORE_CALIBRATED.add(idx)
        try:
            scratch = torch.empty(lib.cnm_calibrate_sweep_store_floats(), device="cuda:%d" % idx, dtype=torch.float32)
        except torch.cuda.OutOfMemoryError:                              # no 630 MB to spare: the default policy (non-temporal) stays, nothing else changes
            return pol
        rc = lib.cnm_calibrate_sweep_store(_p(scratch), scratch.numel(), _stream(), None)
        del scratch
        if rc < 0:
            _lib.check(rc)
        return lib.cnm_tune_sweep_store(99, None)


_SWEEP_STORE_IN_STEP = set()


def calibrate_sweep_store_in_step(step_fn, device=None, n=6, force=False):
    """The same decision measured where it matters: `step_fn()` runs ONE real step of the caller (containing one plane-sweep launch); both
    policies are forced in turn, the launch is timed between its real neighbours by the library's measurement hook, and the policy with
    the lower median becomes the device's (cnm_decide_sweep_store) -- what the launch's stores meet depends on the kernels around it, and
    the scratch calibration's margin is ~1 us.  Blocking, once per device and process, never under capture; skipped when CNM_SWEEP_STORE
    or cnm_tune_sweep_store already forces a policy.  Returns the policy in force."""
    import ctypes
    lib = _lib.load()
    dev = torch.device("cuda" if device is None else device)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    with torch.cuda.device(idx):
        if (idx in _SWEEP_STORE_IN_STEP and not force) or torch.cuda.is_current_stream_capturing() or os.environ.get("CNM_SWEEP_STORE", "") in ("0", "2", "plain", "nt"):
            return lib.cnm_tune_sweep_store(99, None)
        before = lib.cnm_tune_sweep_store(99, None)
        med_prev = (ctypes.c_float * 2)()
        lib.cnm_tune_sweep_store(99, ctypes.cast(med_prev, ctypes.c_void_p))
        _SWEEP_STORE_IN_STEP.add(idx)
        med = (ctypes.c_float * 2)()
        buf = (ctypes.c_float * n)()
        try:
            for k, pol in enumerate((0, 2)):
                lib.cnm_tune_sweep_store(pol, None)
                for _ in range(2):
                    step_fn()
                _lib.check(lib.cnm_debug_sweep_timing_arm(n))
                for _ in range(n):
                    step_fn()
                got = lib.cnm_debug_sweep_timing_read(ctypes.cast(buf, ctypes.c_void_p), n)
                lib.cnm_debug_sweep_timing_arm(0)
                torch.cuda.synchronize()
                if got != n:                                             # the step launched no (or several differently many) sweeps: leave the decision as it was
                    return before
                med[k] = sorted(buf[i] for i in range(n))[n // 2] * 1e3
        finally:
            lib.cnm_tune_sweep_store(-1, None)                           # drop the forced policy (this also drops the scratch calibration: restored or replaced below)
            if before >= 0:
                lib.cnm_decide_sweep_store(before, ctypes.cast(med_prev, ctypes.c_void_p))
        _lib.check(lib.cnm_decide_sweep_store(0 if med[0] < med[1] else 2, ctypes.cast(med, ctypes.c_void_p)))
        return lib.cnm_tune_sweep_store(99, None)


def conv3x3_winograd4_c4(x, u_packed, b_packed, Cout, relu=True, x2=None, ksize=3, sync=None):
    """36-point Winograd twin of conv2d_c4(stride=1): ksize 3 -> F(4x4,3x3), ksize 5 -> F(2x2,5x5).  sync = a
    wino36_sync_workspace lets the staged 3x3 kernel split its work evenly over the CUs (cnm_conv3x3_winograd4_sync_c4_f32)."""
    _dev(x, u_packed, b_packed, x2, sync)
    N, G, H, W, _ = x.shape
    out = torch.empty(N, Cout // 4, H, W, 4, device=x.device, dtype=torch.float32)
    G2 = x2.shape[1] if x2 is not None else 0
    lib = _lib.load()
    with torch.cuda.device(x.device):
        if sync is not None:
            fn = lib.cnm_conv3x3_winograd4_sync_c4_f32 if ksize == 3 else lib.cnm_conv5x5_winograd_sync_c4_f32
            _lib.check(fn(_p(x), G, 0, G, _p(x2) if x2 is not None else None, G2, 0, G2,
                          _p(out), Cout // 4, 0, Cout, _p(u_packed), _p(b_packed), N, H, W, int(relu), _p(sync), sync.numel(), _stream()))
        else:
            fn = lib.cnm_conv3x3_winograd4_c4_f32 if ksize == 3 else lib.cnm_conv5x5_winograd_c4_f32
            _lib.check(fn(_p(x), G, 0, G, _p(x2) if x2 is not None else None, G2, 0, G2,
                          _p(out), Cout // 4, 0, Cout, _p(u_packed), _p(b_packed), N, H, W, int(relu), _stream()))
    return out


def repack_winograd4_quad(u_packed, Cout, Cin):
    """The 36-point packed filter of pack_winograd4 (3x3) re-ordered for the four-wave kernel's 8-channel phases (a permutation)."""
    _dev(u_packed)
    lib = _lib.load()
    uq = torch.empty(lib.cnm_packed_winograd4_quad_floats(Cout, Cin), device=u_packed.device, dtype=torch.float32)
    with torch.cuda.device(u_packed.device):
        _lib.check(lib.cnm_repack_winograd4_quad_f32(_p(u_packed), Cout, Cin, _p(uq), _stream()))
    return uq


def conv3x3_winograd4q_c4(x, uq_packed, b_packed, Cout, relu=True, x2=None, sync=None):
    """Four-wave F(4x4,3x3) kernel (cnm_conv3x3_winograd4q_sync_c4_f32): 64 output channels x 32 tiles per workgroup, one wave per SIMD."""
    _dev(x, uq_packed, b_packed, x2, sync)
    N, G, H, W, _ = x.shape
    out = torch.empty(N, Cout // 4, H, W, 4, device=x.device, dtype=torch.float32)
    G2 = x2.shape[1] if x2 is not None else 0
    lib = _lib.load()
    with torch.cuda.device(x.device):
        _lib.check(lib.cnm_conv3x3_winograd4q_sync_c4_f32(_p(x), G, 0, G, _p(x2) if x2 is not None else None, G2, 0, G2,
                                                          _p(out), Cout // 4, 0, Cout, _p(uq_packed), _p(b_packed), N, H, W, int(relu),
                                                          _p(sync) if sync is not None else None, sync.numel() if sync is not None else 0, _stream()))
    return out


# rows of the bilinear 2x upsampling (align_corners=False) seen by a 3-tap window, as weights on low-resolution rows
# (i-1, i, i+1): [output parity a][tap k] -- hi-res row 2i+a+k-1
_UPS_TAPS = (((0.75, 0.25, 0.0), (0.25, 0.75, 0.0), (0.0, 0.75, 0.25)),
             ((0.25, 0.75, 0.0), (0.0, 0.75, 0.25), (0.0, 0.25, 0.75)))


def compose_upsample_filters(weight):
    """3x3 filter applied AFTER nn.Upsample(scale_factor=2, mode='bilinear') == four 3x3 filters (one per output row /
    column parity) applied to the low-resolution image: [Cout,Cin,3,3] -> [4*Cout,Cin,3,3], phase (2a+b) major.
    Exact away from the image border (the composition corresponds to replicate padding of the upsampled image)."""
    U = torch.tensor(_UPS_TAPS, dtype=torch.float64, device=weight.device)
    wv = torch.einsum("oikl,akp,blq->aboipq", weight.double(), U, U)
    return wv.reshape(4 * weight.shape[0], weight.shape[1], 3, 3).float().contiguous()


def pack_winograd4_upsampled(weight, bn=None, eps=1e-5):
    """(u_packed, b_packed, w_ring) for conv3x3_upsampled_winograd4_c4: the composed phase filters as 4*Cout output
    channels, the folded bias four times, and the plain filter in MFMA order for the ring pass."""
    _dev(weight, *(bn or ()))
    lib = _lib.load()
    Cout, Cin = weight.shape[:2]
    rep = tuple(t.repeat(4) for t in bn) if bn else None
    wr = torch.empty(lib.cnm_packed_upsampled_ring_floats(Cout, Cin), device=weight.device, dtype=torch.float32)
    g, v = (_c(bn[0]), _c(bn[3])) if bn else (None, None)
    with torch.cuda.device(weight.device):
        _lib.check(lib.cnm_pack_upsampled_ring_f32(_p(_c(weight)), _p(g), _p(v), eps, Cout, Cin, _p(wr), _stream()))
    return pack_winograd4(compose_upsample_filters(weight), rep, 0, eps), pack_conv(weight, bn, eps=eps)[1].repeat(4).contiguous(), wr


def conv3x3_upsampled_winograd4_c4(x, u_packed, b_packed, Cout, relu=True, w_ring=None, out=None, sync=None):
    """conv3x3(upsample2x(x)) + bias (+ ReLU) without materialising the upsampled tensor: x [N,G,H,W,4] -> [N,Cout/4,2H,2W,4].
    With w_ring (pack_winograd4_upsampled) the ring pass follows and the result is the reference's zero-padded one;
    without it the upsampled image is replicate-padded (differs on the one-pixel output ring)."""
    _dev(x, u_packed, b_packed, w_ring)
    N, G, H, W, _ = x.shape
    if out is None:
        out = torch.empty(N, Cout // 4, 2 * H, 2 * W, 4, device=x.device, dtype=torch.float32)
    lib = _lib.load()
    with torch.cuda.device(x.device):
        if sync is not None:
            _lib.check(lib.cnm_conv3x3_upsampled_winograd4_sync_c4_f32(_p(x), G, 0, G, _p(out), Cout // 4, 0, Cout, _p(u_packed), _p(b_packed),
                                                                       N, H, W, int(relu), int(w_ring is not None), _p(sync), sync.numel(), _stream()))
        else:
            _lib.check(lib.cnm_conv3x3_upsampled_winograd4_c4_f32(_p(x), G, 0, G, _p(out), Cout // 4, 0, Cout, _p(u_packed), _p(b_packed),
                                                                  N, H, W, int(relu), int(w_ring is not None), _stream()))
        if w_ring is not None:
            _lib.check(lib.cnm_conv3x3_upsampled_ring_c4_f32(_p(x), G, 0, G, _p(out), Cout // 4, 0, Cout, _p(w_ring), _p(b_packed),
                                                             N, H, W, int(relu), _stream()))
    return out


def pack_upsampled_f16(weight, bn=None, eps=1e-5):
    """(w_packed half, b_packed, w_ring) for conv3x3_upsampled_c8: the composed phase filters as 4*Cout output channels on
    the fp16 kernel, the folded bias four times, and the plain filter in MFMA order (fp32) for the ring pass."""
    _dev(weight, *(bn or ()))
    lib = _lib.load()
    Cout, Cin = weight.shape[:2]
    rep = tuple(t.repeat(4) for t in bn) if bn else None
    wr = torch.empty(lib.cnm_packed_upsampled_ring_floats(Cout, Cin), device=weight.device, dtype=torch.float32)
    g, v = (_c(bn[0]), _c(bn[3])) if bn else (None, None)
    with torch.cuda.device(weight.device):
        _lib.check(lib.cnm_pack_upsampled_ring_f32(_p(_c(weight)), _p(g), _p(v), eps, Cout, Cin, _p(wr), _stream()))
    wp, bp = pack_conv_f16(compose_upsample_filters(weight), rep, None, 0, eps)
    return wp, bp, wr


def conv3x3_upsampled_c8(x, w_packed, b_packed, Cout, relu=True, w_ring=None):
    """fp16 conv3x3(upsample2x(x)) + bias (+ ReLU): x [N,G,H,W,8] half -> [N,Cout/8,2H,2W,8] half; with w_ring the
    reference's zero-padded result, without it replicate padding of the upsampled image (differs on the output ring)."""
    x = x.contiguous()
    N, G, H, W, _ = x.shape
    out = torch.empty(N, Cout // 8, 2 * H, 2 * W, 8, device=x.device, dtype=torch.float16)
    lib = _lib.load()
    with torch.cuda.device(x.device):
        _lib.check(lib.cnm_conv3x3_upsampled_c8_f16(x.data_ptr(), G, 0, G, out.data_ptr(), Cout // 8, 0, Cout, w_packed.data_ptr(), _p(b_packed),
                                                    N, H, W, int(relu), int(w_ring is not None), _stream()))
        if w_ring is not None:
            _lib.check(lib.cnm_conv3x3_upsampled_ring_c8_f16(x.data_ptr(), G, 0, G, out.data_ptr(), Cout // 8, 0, Cout, _p(w_ring), _p(b_packed),
                                                             N, H, W, int(relu), _stream()))
    return out


def conv_rows_winograd_c4(x, u_packed, b_packed, Cout, ksize, relu=True, x2=None, stride=1, tile=None, sync=None):
    """Row-wise Winograd twin of conv2d_c4(ksize=5|7, stride=1|2).  sync: optional wino36_sync_workspace() for the staged
    7x7 stride-1 kernel (equal shares of the reduction per CU)."""
    _dev(x, u_packed, b_packed, x2)
    N, G, H, W, _ = x.shape
    pad = ksize // 2
    tile = rows_tile(ksize, stride) if tile is None else tile
    Ho, Wo = (H + 2 * pad - ksize) // stride + 1, (W + 2 * pad - ksize) // stride + 1
    out = torch.empty(N, Cout // 4, Ho, Wo, 4, device=x.device, dtype=torch.float32)
    G2 = x2.shape[1] if x2 is not None else 0
    with torch.cuda.device(x.device):
        _lib.check(_lib.load().cnm_conv_rows_winograd_sync_c4_f32(_p(x), G, 0, G, _p(x2) if x2 is not None else None, G2, 0, G2,
                                                                  _p(out), Cout // 4, 0, Cout, _p(u_packed), _p(b_packed),
                                                                  N, H, W, ksize, stride, tile, int(relu),
                                                                  _p(sync), sync.numel() if sync is not None else 0, _stream()))
    return out


def conv3x3_winograd_c4(x, u_packed, b_packed, Cout, relu=True, x2=None):
    """Winograd F(2x2,3x3) twin of conv2d_c4(ksize=3, stride=1)."""
    _dev(x, u_packed, b_packed, x2)
    N, G, H, W, _ = x.shape
    out = torch.empty(N, Cout // 4, H, W, 4, device=x.device, dtype=torch.float32)
    G2 = x2.shape[1] if x2 is not None else 0
    with torch.cuda.device(x.device):
        _lib.check(_lib.load().cnm_conv3x3_winograd_c4_f32(_p(x), G, 0, G, _p(x2) if x2 is not None else None, G2, 0, G2,
                                                           _p(out), Cout // 4, 0, Cout, _p(u_packed), _p(b_packed),
                                                           N, H, W, int(relu), _stream()))
    return out


def upsample2x_c4(x):
    _dev(x)
    N, G, H, W, _ = x.shape
    out = torch.empty(N, G, 2 * H, 2 * W, 4, device=x.device, dtype=torch.float32)
    with torch.cuda.device(x.device):
        _lib.check(_lib.load().cnm_upsample2x_c4_f32(_p(x), G, 0, _p(out), G, 0, N, G, H, W, _stream()))
    return out


def upsample2x_c8(x):
    """fp16 c8 [N,G,H,W,8] -> [N,G,2H,2W,8]: nn.Upsample(scale_factor=2, mode='bilinear', align_corners=False)."""
    x = x.contiguous()
    N, G, H, W, _ = x.shape
    out = torch.empty(N, G, 2 * H, 2 * W, 8, device=x.device, dtype=torch.float16)
    with torch.cuda.device(x.device):
        _lib.check(_lib.load().cnm_upsample2x_c8_f16(x.data_ptr(), G, 0, out.data_ptr(), G, 0, N, G, H, W, _stream()))
    return out


def head_sigmoid_c4(x, w_head, bias, scale, up_out=None, up_group=0):
    _dev(x, w_head, bias, up_out)
    N, G, H, W, _ = x.shape
    disp = torch.empty(N, 1, H, W, device=x.device, dtype=torch.float32)
    with torch.cuda.device(x.device):
        _lib.check(_lib.load().cnm_head_sigmoid_c4_f32(_p(x), G, 0, 4 * G, _p(w_head), _p(bias), float(scale), _p(disp),
                                                       _p(up_out), up_out.shape[1] if up_out is not None else 0, up_group,
                                                       N, H, W, _stream()))
    return disp


def head_sigmoid_c8(x, w_head, bias, scale, up_out=None, up_group=0):
    """fp16 twin of head_sigmoid_c4: x [N,G,H,W,8] half, w_head from pack_head (fp32) -> disp [N,1,H,W] fp32."""
    x = x.contiguous()
    N, G, H, W, _ = x.shape
    disp = torch.empty(N, 1, H, W, device=x.device, dtype=torch.float32)
    with torch.cuda.device(x.device):
        _lib.check(_lib.load().cnm_head_sigmoid_c8_f16(x.data_ptr(), G, 0, 8 * G, _p(w_head), _p(bias), float(scale), _p(disp),
                                                       up_out.data_ptr() if up_out is not None else None,
                                                       up_out.shape[1] if up_out is not None else 0, up_group, N, H, W, _stream()))
    return disp


def depth2normal(depth, intrinsic_inv, k_size=9, input_is_idepth=False):
    """depth [B,H,W], K^-1 [B,3,3] -> (normal [B,3,H,W], points [B,3,H,W])   (depth_util.py:149-203)"""
    if host.is_host(depth, intrinsic_inv):
        return host.depth2normal(depth, intrinsic_inv, k_size, input_is_idepth)
    _dev(depth, intrinsic_inv)
    depth, intrinsic_inv = _c(depth), _c(intrinsic_inv)
    B, H, W = depth.shape
    normal = torch.empty(B, 3, H, W, device=depth.device, dtype=torch.float32)
    points = torch.empty_like(normal)
    with torch.cuda.device(depth.device):
        _lib.check(_lib.load().cnm_depth2normal_f32(_p(depth), _p(intrinsic_inv), _p(normal), _p(points),
                                                    B, H, W, k_size, int(input_is_idepth), _stream()))
    return normal, points


def plane_normals(normal, instance_segs, planes_num, with_loss=True):
    """Plane-instance regularisation (reference depth_util.py:205-238 / :243-278): normal [B,3,H,W] float32,
    instance_segs [B,P,H,W] bool, planes_num [B] ints -> (regularised normal (new tensor), loss scalar tensor or None)."""
    _dev(normal)
    B, _, H, W = normal.shape
    P = instance_segs.shape[1]
    pn = torch.as_tensor(planes_num, dtype=torch.int32)
    pmax = int(pn.max()) if pn.numel() else 0
    if pn.numel() != B or pmax > P or int(pn.min()) < 0:
        raise ValueError("planes_num must hold B counts in [0, %d]" % P)
    out = normal.contiguous().clone()
    seg = instance_segs.to(device=normal.device, dtype=torch.uint8).contiguous()
    pn = pn.to(normal.device)
    terms = torch.zeros(B, P, device=normal.device, dtype=torch.float32) if with_loss else None
    with torch.cuda.device(normal.device):
        _lib.check(_lib.load().cnm_plane_normals_f32(_p(out), seg.data_ptr(), pn.data_ptr(), pmax, _p(terms), B, P, H, W, _stream()))
    return out, (terms.sum() if with_loss else None)


def intrinsics_inverse(cam):
    """cam [B,2,4,4] (any batch stride) -> K^-1 [B,3,3]   (train.py:201-202, eval.py:271)"""
    if host.is_host(cam):
        return host.intrinsics_inverse(cam)
    _dev(cam)
    if cam.stride()[-3:] != (16, 4, 1):
        cam = cam.contiguous()
    B = cam.shape[0]
    out = torch.empty(B, 3, 3, device=cam.device, dtype=torch.float32)
    with torch.cuda.device(cam.device):
        _lib.check(_lib.load().cnm_intrinsics_inverse_f32(_p(cam), cam.stride(0) if B > 1 else 32, _p(out), B, _stream()))
    return out


PADDING_MODES = {"zeros": 0, "border": 1, "reflection": 2}           # torch.nn.functional.grid_sample's padding_mode values


def inverse_warp(feat, depth, pose, intrinsics, intrinsics_inv, padding_mode="zeros"):
    if host.is_host(feat, depth, pose, intrinsics, intrinsics_inv):
        return host.inverse_warp(feat, depth, pose, intrinsics, intrinsics_inv, padding_mode)
    _dev(feat, depth, pose, intrinsics, intrinsics_inv)
    if padding_mode not in PADDING_MODES:
        raise ValueError("padding_mode must be one of %s, got %r" % (sorted(PADDING_MODES), padding_mode))
    feat, depth, pose, intrinsics, intrinsics_inv = map(_c, (feat, depth, pose, intrinsics, intrinsics_inv))
    B, Cc, H, W = feat.shape
    out = torch.empty_like(feat)
    with torch.cuda.device(feat.device):
        _lib.check(_lib.load().cnm_inverse_warp_pad_f32(_p(feat), _p(depth), _p(pose), _p(intrinsics), _p(intrinsics_inv),
                                                        _p(out), B, Cc, H, W, PADDING_MODES[padding_mode], _stream()))
    return out

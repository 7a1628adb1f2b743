"""Host twins of the operator API: CPU torch tensors in, CPU torch tensors out, through the `_cpu` entry points of
libcnm_engine.so (csrc/host_twins.cpp, EngHost in csrc/nets.hip).

BASELINE configs[0] ("DepthNet eval, 1 ref + 1 src, 256x192, 32 planes, batch=1 on CPU: plumbing, no GPU") and SURVEY 8(b)'s
"_cpu twin taking host pointers": `cnmnet_amd.ops` and the modules of `cnmnet_amd.depthnet` route CPU tensors here in eval
mode.  Inference only, fp32 only, no oracle: this is the product's own C++ (direct convolutions, one thread per core), written
for plumbing, not speed.
"""
import torch

from . import _lib


def is_host(*tensors):
    ts = [t for t in tensors if t is not None]
    return bool(ts) and all((not t.is_cuda) for t in ts)


def _f32(*tensors):
    for t in tensors:
        if t is not None and (t.is_cuda or t.dtype != torch.float32):
            raise _lib.EngineError("host twins take float32 CPU tensors; got %s on %s" % (t.dtype, t.device))


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


def _p(t):
    return 0 if t is None else t.data_ptr()


def homography_terms(ref_cam, src_cam):
    _f32(ref_cam, src_cam)
    ref_cam, src_cam = _c(ref_cam), _c(src_cam)
    B, S = src_cam.shape[0], src_cam.shape[1]
    out = torch.empty(B * S, 12)
    _lib.check(_lib.load().cnm_homography_terms_cpu(_p(ref_cam), _p(src_cam), _p(out), B, S))
    return out


def plane_sweep_volume_hmkt(left, right, hmkt, lo, hi, planes):
    _f32(left, right, hmkt)
    left, right, hmkt = _c(left), _c(right), _c(hmkt)
    B, _, H, W = left.shape
    vol = torch.empty(B, planes, H, W)
    _lib.check(_lib.load().cnm_planesweep_volume_nchw_cpu(_p(left), _p(right), _p(hmkt), _p(vol), B, 1, H, W, planes, lo, hi))
    return vol


def plane_sweep_cat_c4(ref, src, hmkt, lo, hi, planes):
    _f32(ref, src, hmkt)
    ref, src, hmkt = _c(ref), _c(src), _c(hmkt)
    B, S, _, H, W = src.shape
    x = torch.empty(B * S, planes // 4 + 1, H, W, 4)
    _lib.check(_lib.load().cnm_planesweep_cat_c4_cpu(_p(ref), _p(src), _p(hmkt), _p(x), B, S, H, W, planes, lo, hi))
    return x


def nchw_to_c4(x):
    _f32(x)
    x = _c(x)
    N, Cc, H, W = x.shape
    G = (Cc + 3) // 4
    out = torch.empty(N, G, H, W, 4)
    _lib.check(_lib.load().cnm_nchw_to_c4_cpu(_p(x), _p(out), G, 0, N, Cc, H, W))
    return out


def c4_to_nchw(x, channels=None):
    _f32(x)
    x = _c(x)
    N, G, H, W, _ = x.shape
    Cc = channels or 4 * G
    out = torch.empty(N, Cc, H, W)
    _lib.check(_lib.load().cnm_c4_to_nchw_cpu(_p(x), G, 0, _p(out), N, Cc, H, W))
    return out


def intrinsics_inverse(cam):
    _f32(cam)
    cam = _c(cam)
    B = cam.shape[0]
    out = torch.empty(B, 3, 3)
    _lib.check(_lib.load().cnm_intrinsics_inverse_cpu(_p(cam), 32, _p(out), B))
    return out


def depth2normal(depth, intrinsic_inv, k_size=9, input_is_idepth=False):
    _f32(depth, intrinsic_inv)
    depth, intrinsic_inv = _c(depth), _c(intrinsic_inv)
    B, H, W = depth.shape
    normal = torch.empty(B, 3, H, W)
    points = torch.empty_like(normal)
    _lib.check(_lib.load().cnm_depth2normal_cpu(_p(depth), _p(intrinsic_inv), _p(normal), _p(points), B, H, W, k_size, int(input_is_idepth)))
    return normal, points


def inverse_warp(feat, depth, pose, intrinsics, intrinsics_inv, padding_mode="zeros"):
    _f32(feat, depth, pose, intrinsics, intrinsics_inv)
    if padding_mode != "zeros":
        raise _lib.EngineError("the host twin of inverse_warp implements padding_mode='zeros' (the reference's default and only use)")
    feat, depth, pose, intrinsics, intrinsics_inv = map(_c, (feat, depth, pose, intrinsics, intrinsics_inv))
    B, Cc, H, W = feat.shape
    out = torch.empty_like(feat)
    _lib.check(_lib.load().cnm_inverse_warp_cpu(_p(feat), _p(depth), _p(pose), _p(intrinsics), _p(intrinsics_inv), _p(out), B, Cc, H, W))
    return out


def pack_conv(weight, bn=None, bias=None, rot=0, eps=1e-5):
    """weight [Cout,Cin,k,k]; bn = (gamma, beta, mean, var) or None -> (w_packed [Cout,k*k,4*ceil(Cin/4)], b_packed [Cout])."""
    _f32(weight, bias, *(bn or ()))
    lib = _lib.load()
    Cout, Cin, k, _ = weight.shape
    wp = torch.empty(lib.cnm_packed_conv_floats_cpu(Cout, Cin, k))
    bp = torch.empty(Cout)
    g, b, m, v = (_c(t.float()) for t in bn) if bn else (None,) * 4
    _lib.check(lib.cnm_pack_conv_bn_cpu(_p(_c(weight)), _p(g), _p(b), _p(m), _p(v), _p(_c(bias)) if bias is not None else 0, float(eps),
                                        Cout, Cin, k, rot, _p(wp), _p(bp)))
    return wp, bp


def pack_head(weight):
    _f32(weight)
    C = weight.shape[1]
    wh = torch.empty(9 * C)
    _lib.check(_lib.load().cnm_pack_head_cpu(_p(_c(weight)), C, _p(wh)))
    return wh


def conv2d_c4(x, w_packed, b_packed, Cout, ksize, stride=1, relu=True, x2=None):
    """Operator-level twin of ops.conv2d_c4 (tests): x [N,G,H,W,4] (+ x2) -> [N,Cout/4,Ho,Wo,4]."""
    _f32(x, x2, w_packed, b_packed)
    x = _c(x)
    N, G, H, W, _ = x.shape
    G2 = 0
    if x2 is not None:
        x2 = _c(x2); G2 = x2.shape[1]
    Ho, Wo = -(-H // stride), -(-W // stride)
    out = torch.empty(N, Cout // 4, Ho, Wo, 4)
    _lib.check(_lib.load().cnm_conv2d_cat2_c4_cpu(_p(x), G, 0, G, _p(x2), G2, 0, G2, _p(out), Cout // 4, 0, Cout, _p(w_packed), _p(b_packed),
                                                  N, H, W, ksize, stride, int(relu)))
    return out


def depthnet_forward(weights_arr, idepth_scale, planes, ref, src, ref_cam, src_cam):
    """depthNet.forward_pairs on the host: -> ([disp1..4], iconv1 c4 [B*S,16,H,W,4])."""
    _f32(ref, src, ref_cam, src_cam)
    lib = _lib.load()
    ref, src, ref_cam, src_cam = (_c(t) for t in (ref, src, ref_cam, src_cam))
    B, S, _, H, W = src.shape
    P = B * S
    disp = [torch.empty(P, 1, H >> i, W >> i) for i in range(4)]
    feat = torch.empty(P, 16, H, W, 4)
    n = lib.cnm_depthnet_workspace_floats_cpu(P, H, W, planes)
    ws = torch.empty(n)
    _lib.check(lib.cnm_depthnet_forward_cpu(weights_arr, float(idepth_scale), planes, _p(ref), _p(src), _p(ref_cam), _p(src_cam),
                                            _p(disp[0]), _p(disp[1]), _p(disp[2]), _p(disp[3]), _p(feat), _p(ws), ws.numel(), B, S, H, W))
    return disp, feat


def refinenet_forward(weights_arr, idepth_scale, idepth01, idepth02, idepth_stride, f1, G1_total, g1, f2, G2_total, g2, N, H, W, return_volume=False):
    _f32(idepth01, idepth02, f1, f2)
    lib = _lib.load()
    disp = torch.empty(N, 1, H, W)
    prob = torch.empty_like(disp)
    vol = torch.empty(N, 16, H, W, 4) if return_volume else None
    ws = torch.empty(lib.cnm_refinenet_workspace_floats_cpu(N, H, W))
    _lib.check(lib.cnm_refinenet_forward_cpu(weights_arr, float(idepth_scale), _p(idepth01), _p(idepth02), idepth_stride,
                                             _p(f1), G1_total, g1, _p(f2), G2_total, g2, _p(disp), _p(prob), _p(vol), _p(ws), ws.numel(), N, H, W))
    return disp, prob, vol


def refinenet_forward_multi(weights_arr, idepth_scale, idepth_pairs, feat_pairs_c4, S, return_volume=False):
    _f32(idepth_pairs, feat_pairs_c4)
    lib = _lib.load()
    idepth_pairs, feat_pairs_c4 = _c(idepth_pairs), _c(feat_pairs_c4)
    P, _, H, W = idepth_pairs.shape
    B = P // S
    disp = torch.empty(B, 1, H, W)
    prob = torch.empty_like(disp)
    vol = torch.empty(B, 16, H, W, 4) if return_volume else None
    ws = torch.empty(lib.cnm_refinenet_workspace_floats_cpu(B, H, W))
    _lib.check(lib.cnm_refinenet_forward_multi_cpu(weights_arr, float(idepth_scale), _p(idepth_pairs), _p(feat_pairs_c4), S,
                                                   _p(disp), _p(prob), _p(vol), _p(ws), ws.numel(), B, H, W))
    return disp, prob, vol

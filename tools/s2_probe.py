"""Stride-2 5x5 / 7x7 layers: the four-pixel-phase form on the staged 36-point kernel (cnm_conv_s2_winograd4_sync_c4_f32) against
the row-wise phase kernel (cnm_conv_rows_winograd_sync_c4_f32), at the bench step's shapes (GPU box only): error of both against
the fp64 torch convolution on a small batch, then time over 20 launches each, interleaved.
   python tools/s2_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cnmnet_amd import ops, _lib

dev = "cuda"
lib = _lib.load()
SYNC = ops.wino36_sync_workspace(dev)


def event_ms(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


torch.manual_seed(0)
for name, k, Cin, Cout, N, H, W in (("depth conv2.3", 5, 256, 256, 16, 96, 128), ("depth conv1.3", 7, 128, 128, 16, 192, 256), ("refine conv3.3", 3, 512, 512, 8, 48, 64), ("depth conv4.3", 3, 512, 512, 16, 24, 32), ("depth conv3.3", 3, 512, 512, 16, 48, 64), ("refine conv1.3", 3, 128, 128, 8, 192, 256), ("refine conv2.3", 3, 256, 256, 8, 96, 128)):
    w = torch.randn(Cout, Cin, k, k, device=dev) * (2.0 / (Cin * k * k)) ** 0.5
    bp = torch.randn(Cout, device=dev) * 0.1
    us = ops.pack_winograd4_s2(w)
    if k == 3:                                                           # 3x3: the row kernel (two F(4,2) column phases) and the implicit GEMM are the alternatives
        ur = ops.pack_winograd_rows(w, stride=2, tile=4); wp, _ = ops.pack_conv(w)
        rows = lambda t: ops.conv_rows_winograd_c4(t, ur, bp, Cout, 3, True, stride=2, tile=4)
        direct = lambda t: ops.conv2d_c4(t, wp, bp, Cout, 3, 2, True)
    else:
        ur = ops.pack_winograd(w, stride=2)
        rows = lambda t: ops.conv_rows_winograd_c4(t, ur, bp, Cout, k, True, stride=2, sync=SYNC)
        direct = None
    xs = torch.randn(2, Cin, H, W, device=dev)
    ref = torch.relu(torch.nn.functional.conv2d(xs.double(), w.double(), bp.double(), stride=2, padding=k // 2)).float()
    xc = ops.nchw_to_c4(xs)
    o_s = ops.c4_to_nchw(ops.conv_s2_winograd4_c4(xc, us, bp, Cout, k, True, sync=SYNC), Cout)
    o_r = ops.c4_to_nchw(rows(xc), Cout)
    print("%s  %dx%d stride 2, %d->%d, %dx%d: |staged phases - torch64| = %.2e   |rows - torch64| = %.2e   (output max %.2f)" % (
        name, k, k, Cin, Cout, H, W, (o_s - ref).abs().max().item(), (o_r - ref).abs().max().item(), ref.abs().max().item()), flush=True)
    x = ops.nchw_to_c4(torch.randn(N, Cin, H, W, device=dev))
    flop = 2.0 * Cout * Cin * k * k * (H // 2) * (W // 2) * N
    for rep in range(2):
        t_r = event_ms(lambda: rows(x))
        t_s = event_ms(lambda: ops.conv_s2_winograd4_c4(x, us, bp, Cout, k, True, sync=SYNC))
        ex_r = ((k + 1) // 2 + 3) / 2.0 / k
        ex_s = (36.0 / 9 if k == 7 else 36.0 / 16) * 4 / (k * k)
        extra = ""
        if direct is not None:
            t_d = event_ms(lambda: direct(x))
            extra = "   implicit GEMM %.3f ms (%.1f TF = %.2f of peak)" % (t_d, flop / t_d / 1e9, flop / t_d / 1e9 / 157.3)
        print("   N%d: rows %.3f ms (executed %.1f TF = %.2f of peak)   staged phases %.3f ms (executed %.1f TF = %.2f of peak)   %.2fx%s" % (
            N, t_r, flop * ex_r / t_r / 1e9, flop * ex_r / t_r / 1e9 / 157.3, t_s, flop * ex_s / t_s / 1e9, flop * ex_s / t_s / 1e9 / 157.3, t_r / t_s, extra), flush=True)
assert ops.sync_workspace_state(SYNC) == 0

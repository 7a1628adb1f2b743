"""fp16 convolution layers at the bench shapes: register-staged kernel against the LDS-DMA kernel (cnm_tune_f16_glds_min_tiles),
with the result of both checked against each other.  GPU box only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cnmnet_amd import _lib, ops

CASES = [  # (N, Cin, Cout, H, W, ksize, stride)
    (16, 67, 128, 192, 256, 7, 1), (16, 128, 128, 192, 256, 7, 2), (16, 128, 256, 96, 128, 5, 1), (16, 256, 256, 96, 128, 5, 2),
    (16, 256, 512, 48, 64, 3, 1), (16, 512, 512, 48, 64, 3, 2), (16, 512, 512, 24, 32, 3, 1), (16, 1024, 512, 24, 32, 3, 1),
    (16, 513, 256, 48, 64, 3, 1), (16, 257, 128, 96, 128, 3, 1), (16, 128, 64, 192, 256, 3, 1), (16, 65, 64, 192, 256, 3, 1),
    (16, 512, 512, 24, 32, 3, 2), (16, 512, 512, 12, 16, 3, 1), (16, 1024, 512, 12, 16, 3, 1), (16, 512, 512, 12, 16, 3, 2), (16, 512, 512, 6, 8, 3, 1),
    (8, 256, 256, 96, 128, 3, 2), (8, 256, 512, 48, 64, 3, 1), (8, 512, 512, 48, 64, 3, 2),
    (8, 67, 128, 192, 256, 3, 1), (8, 128, 128, 192, 256, 3, 2), (8, 512, 256, 48, 64, 3, 1), (8, 256, 128, 96, 128, 3, 1), (8, 64, 64, 192, 256, 3, 1),
]


def bench(fn, iters=10):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


lib = _lib.load(); dev = "cuda"
tot = {0: 0.0, 1: 0.0}
for N, Cin, Cout, H, W, k, st in CASES:
    w = torch.randn(Cout, Cin, k, k, device=dev) * 0.02
    x = ops.nchw_to_c8(torch.randn(N, Cin, H, W, device=dev))
    wp, bp = ops.pack_conv_f16(w, None, torch.randn(Cout, device=dev))
    fn = lambda: ops.conv2d_c8(x, wp, bp, Cout, k, st, True)
    res = {}
    for mode, thr in ((0, 1 << 30), (1, 1)):
        lib.cnm_tune_f16_glds_min_tiles(thr)
        y = fn(); torch.cuda.synchronize()
        res[mode] = (bench(fn), y.float())
        tot[mode] += res[mode][0]
    diff = (res[0][1] - res[1][1]).abs().max().item()
    gf = 2.0 * Cout * Cin * k * k * (H // st) * (W // st) * N / 1e9
    print("N%2d %4d->%4d k%d s%d %3dx%-3d: staged %.3f ms (%4.0f TF)  glds %.3f ms (%4.0f TF)  %+.1f %%  maxdiff %.2e (max %.1f)" % (
        N, Cin, Cout, k, st, H, W, res[0][0], gf / res[0][0], res[1][0], gf / res[1][0], 100 * (res[0][0] / res[1][0] - 1), diff, res[0][1].abs().max().item()))
print("total staged %.2f ms, glds %.2f ms" % (tot[0], tot[1]))

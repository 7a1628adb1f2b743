"""Relative L2 error (the training tests' metric) of the 5x5 stride-1 kernels against the fp64 convolution: rows F(2,5) vs 36-point F(2x2,5x5) (GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cnmnet_amd import ops
dev = "cuda"
sync = ops.wino36_sync_workspace(dev)
def rel(a, b): return float((a.double() - b).norm() / b.norm())
torch.manual_seed(1)
for cin, cout, N, H, W in ((128, 256, 2, 48, 64), (128, 256, 8, 96, 128), (256, 128, 2, 48, 64)):
    x = torch.randn(N, cin, H, W, device=dev); w = torch.randn(cout, cin, 5, 5, device=dev) * 0.05
    ref = torch.nn.functional.conv2d(x.double(), w.double(), padding=2)
    xc = ops.nchw_to_c4(x)
    r2 = ops.c4_to_nchw(ops.conv_rows_winograd_c4(xc, ops.pack_winograd(w, tile=2), None, cout, 5, relu=False, tile=2), cout)
    r36 = ops.c4_to_nchw(ops.conv3x3_winograd4_c4(xc, ops.pack_winograd4(w), None, cout, relu=False, ksize=5, sync=sync), cout)
    d = torch.nn.functional.conv2d(x, w, padding=2)
    w3 = w[:, :, 1:4, 1:4].contiguous(); ref3 = torch.nn.functional.conv2d(x.double(), w3.double(), padding=1)
    r4 = ops.c4_to_nchw(ops.conv3x3_winograd4_c4(xc, ops.pack_winograd4(w3), None, cout, relu=False, sync=sync), cout)
    print("%d->%d N%d %dx%d: rel L2 vs fp64: rows F(2,5) %.2e | F(2x2,5x5) %.2e | torch fp32 %.2e | (3x3 F(4x4,3x3) %.2e)" % (cin, cout, N, H, W, rel(r2, ref), rel(r36, ref), rel(d, ref), rel(r4, ref3)))
for cin, cout, N, H, W in ((67, 128, 2, 48, 64), (67, 128, 4, 192, 256)):
    x = torch.randn(N, cin, H, W, device=dev).abs(); w = torch.randn(cout, cin, 7, 7, device=dev) * 0.03
    ref = torch.nn.functional.conv2d(x.double(), w.double(), padding=3)
    xc = ops.nchw_to_c4(x)
    r2 = ops.c4_to_nchw(ops.conv_rows_winograd_c4(xc, ops.pack_winograd(w, tile=2), None, cout, 7, relu=False, tile=2), cout)
    r4 = ops.c4_to_nchw(ops.conv_rows_winograd_c4(xc, ops.pack_winograd(w, tile=4), None, cout, 7, relu=False, tile=4, sync=sync), cout)
    print("7x7 %d->%d N%d %dx%d: rel L2 vs fp64: rows F(2,7) %.2e | F(4,7) %.2e" % (cin, cout, N, H, W, rel(r2, ref), rel(r4, ref)))
    import time
    for tile in (2, 4):
        up = ops.pack_winograd(w, tile=tile)
        f = lambda: ops.conv_rows_winograd_c4(xc, up, None, cout, 7, relu=False, tile=tile, sync=sync)
        for _ in range(3): f()
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(10): f()
        torch.cuda.synchronize(); print("   tile %d: %.3f ms" % (tile, (time.perf_counter() - t) * 100))

"""Winograd conv vs the direct implicit-GEMM conv on the same inputs: error + timing (GPU)."""
import sys, torch
sys.path.insert(0, ".")
from cnmnet_amd import ops

def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n

torch.manual_seed(0)
dev = "cuda"
import os
K = int(os.environ.get("K", "3")); ST = int(os.environ.get("ST", "1")); ALG = int(os.environ.get("ALG", "2"))
cases = [(16, 128, 64, 192, 256), (16, 64, 64, 192, 256), (16, 256, 128, 96, 128), (16, 512, 256, 48, 64), (16, 512, 512, 12, 16),
         (16, 1024, 512, 12, 16), (16, 512, 512, 6, 8), (16, 65, 64, 96, 128), (2, 67, 64, 10, 14), (16, 32, 64, 384, 512)]
if K != 3:
    cases = [(16, 67, 128, 192, 256), (16, 128, 256, 96, 128), (2, 67, 128, 23, 31), (1, 128, 64, 5, 9), (16, 68, 128, 96, 128)]
for (N, Cin, Cout, H, W) in cases:
    w = torch.randn(Cout, Cin, K, K, device=dev) * (2.0 / (K * K * Cin)) ** 0.5
    bn = (torch.rand(Cout, device=dev) + 0.5, torch.randn(Cout, device=dev) * 0.1, torch.randn(Cout, device=dev) * 0.1, torch.rand(Cout, device=dev) + 0.5)
    x = torch.randn(N, Cin, H, W, device=dev)
    xc = ops.nchw_to_c4(x)
    wp, bp = ops.pack_conv(w, bn=bn)
    up = ops.pack_winograd4(w, bn=bn) if (K in (3, 5) and ALG == 4) else ops.pack_winograd(w, bn=bn, stride=ST)
    y0 = ops.conv2d_c4(xc, wp, bp, Cout, K, ST, True)
    y1 = (ops.conv3x3_winograd4_c4(xc, up, bp, Cout, True, ksize=K) if (ALG == 4 and K in (3, 5)) else (ops.conv3x3_winograd_c4(xc, up, bp, Cout, True) if K == 3 else ops.conv_rows_winograd_c4(xc, up, bp, Cout, K, True, stride=ST)))
    ref = torch.nn.functional.conv2d(x.double(), w.double(), stride=ST, padding=K // 2)
    sc = (bn[0] / torch.sqrt(bn[3] + 1e-5)).double()
    ref = torch.relu(ref * sc[None, :, None, None] + (bn[1].double() - bn[2].double() * sc)[None, :, None, None])
    e0 = (ops.c4_to_nchw(y0).double() - ref).abs().max().item()
    e1 = (ops.c4_to_nchw(y1).double() - ref).abs().max().item()
    t0 = timeit(lambda: ops.conv2d_c4(xc, wp, bp, Cout, K, ST, True))
    t1 = timeit(lambda: (ops.conv3x3_winograd4_c4(xc, up, bp, Cout, True, ksize=K) if (ALG == 4 and K in (3, 5)) else (ops.conv3x3_winograd_c4(xc, up, bp, Cout, True) if K == 3 else ops.conv_rows_winograd_c4(xc, up, bp, Cout, K, True, stride=ST))))
    fl = 2.0 * N * (H // ST) * (W // ST) * Cout * Cin * K * K
    print("N%d %4d->%3d %3dx%3d  direct err %.2e %.3f ms %.0f TF | wino err %.2e %.3f ms %.0f TF(eff)  x%.2f" %
          (N, Cin, Cout, H, W, e0, t0, fl / t0 / 1e9, e1, t1, fl / t1 / 1e9, t0 / t1), flush=True)

"""The 21 F(4x4,3x3) launches of a bench step (16 depthNet pairs + 8 refine frames, 192x256), each timed alone: executed
TFLOP/s (36/144 of the direct flops) against the 157.3 TF fp32-MFMA peak.  GPU box only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cnmnet_amd import ops
LAYERS = [  # (name, N, Cin, Cout, H, W)
    ("depth conv3.0", 16, 256, 512, 48, 64), ("depth conv4.0", 16, 512, 512, 24, 32), ("depth upconv3", 16, 512, 256, 48, 64), ("depth iconv3", 16, 513, 256, 48, 64),
    ("depth iconv2", 16, 257, 128, 96, 128), ("depth iconv1", 16, 65, 64, 192, 256), ("depth iconv4", 16, 1024, 512, 24, 32), ("depth upconv4", 16, 512, 512, 24, 32),
    ("refine conv1.0", 8, 67, 128, 192, 256), ("refine conv2.0", 8, 128, 256, 96, 128), ("refine conv3.0", 8, 256, 512, 48, 64),
    ("refine upconv3 x2", 8, 512, 256, 48, 64), ("refine iconv3 x2", 8, 512, 256, 48, 64), ("refine upconv2 x2", 8, 256, 128, 96, 128),
    ("refine iconv2 x2", 8, 256, 128, 96, 128), ("refine iconv1 x2", 8, 64, 64, 192, 256)]
def bench(fn, iters=10):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
dev = "cuda"; tot = 0
for name, N, Cin, Cout, H, W in LAYERS:
    x = ops.nchw_to_c4(torch.randn(N, Cin, H, W, device=dev)); up = ops.pack_winograd4(torch.randn(Cout, Cin, 3, 3, device=dev) * 0.02); bp = torch.zeros(Cout, device=dev)
    ms = bench(lambda: ops.conv3x3_winograd4_c4(x, up, bp, Cout, True))
    gf = 2.0 * Cout * Cin * 9 * H * W * N / 1e9
    wgs = (Cout // 64) * -(-(N * -(-H // 4) * -(-W // 4)) // 16)
    print("%-20s %4d->%4d %3dx%-3d N%2d: %.3f ms  executed %5.1f TF = %.2f of peak   (%d workgroups = %.2f rounds of 512, %d chunks)" % (
        name, Cin, Cout, H, W, N, ms, gf / ms / 4, gf / ms / 4 / 157.3, wgs, wgs / 512, -(-(4 * -(-Cin // 4)) // 16)))

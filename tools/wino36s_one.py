"""One F(4x4,3x3) layer in a loop (for rocprofv3 / ablation timing):  python tools/wino36s_one.py Cin Cout H W N [staged 0|1] [ablate mask] [iters]"""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cnmnet_amd import ops, _lib
Cin, Cout, H, W, N = [int(v) for v in sys.argv[1:6]]
staged = int(sys.argv[6]) if len(sys.argv) > 6 else 1
abl = int(sys.argv[7]) if len(sys.argv) > 7 else 0
iters = int(sys.argv[8]) if len(sys.argv) > 8 else 30
lib = _lib.load(); lib.cnm_tune_wino36_staged(2 if staged else 0)
if abl:
    f = ctypes.CDLL(_lib.LIB_PATH).cnm_tune_wino36s_ablate; f.argtypes = [ctypes.c_int]; f(abl)
x = ops.nchw_to_c4(torch.randn(N, Cin, H, W, device="cuda")); up = ops.pack_winograd4(torch.randn(Cout, Cin, 3, 3, device="cuda") * 0.02); bp = torch.zeros(Cout, device="cuda")
fn = lambda: ops.conv3x3_winograd4_c4(x, up, bp, Cout, True)
for _ in range(3): fn()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters): fn()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / iters
gf = 2.0 * Cout * Cin * 9 * H * W * N / 1e9
print("%d->%d %dx%d N%d staged=%d ablate=%2d: %.4f ms  %.1f TF executed (%.3f of 157.3)" % (Cin, Cout, H, W, N, staged, abl, ms, gf / ms / 4, gf / ms / 4 / 157.3), flush=True)

"""One F(4x4,3x3) layer on the four-wave kernel in a loop (ablation timing / rocprofv3):
   python tools/wino36q_one.py Cin Cout H W N [ablate mask] [iters]     (CNM_ENGINE_LIB = an -DWINO4Q_ABLATE build for masks != 0)"""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cnmnet_amd import ops, _lib
Cin, Cout, H, W, N = [int(v) for v in sys.argv[1:6]]
abl = int(sys.argv[6]) if len(sys.argv) > 6 else 0
iters = int(sys.argv[7]) if len(sys.argv) > 7 else 30
lib = _lib.load()
if abl:
    f = ctypes.CDLL(_lib.LIB_PATH).cnm_tune_wino36q_ablate; f.argtypes = [ctypes.c_int]; f(abl)
ct = 4 * ((Cin + 3) // 4)
x = ops.nchw_to_c4(torch.randn(N, Cin, H, W, device="cuda"))
uq = ops.repack_winograd4_quad(ops.pack_winograd4(torch.randn(Cout, Cin, 3, 3, device="cuda") * 0.02), Cout, Cin); bp = torch.zeros(Cout, device="cuda")
sync = ops.wino36_sync_workspace("cuda")
fn = lambda: ops.conv3x3_winograd4q_c4(x, uq, bp, Cout, True, sync=sync)
for _ in range(3): fn()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters): fn()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / iters
gf = 2.0 * Cout * (8 * ((ct // 4 + 1) // 2)) * 36 * ((H + 3) // 4) * ((W + 3) // 4) * N / 1e9
print("%d->%d %dx%d N%d quad ablate=%2d: %.4f ms  %.1f TF executed (%.3f of 157.3)" % (Cin, Cout, H, W, N, abl, ms, gf / ms, gf / ms / 157.3), flush=True)
if abl:
    ops.engine_status(clear=True)

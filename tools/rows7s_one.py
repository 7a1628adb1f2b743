"""Time one conv1.0-shaped launch of the staged 7x7 rows kernel: rows7s_one.py [ablate-mask] [sync 0|1] [N H W Cin]."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cnmnet_amd import ops, _lib
lib = _lib.load(); dev = "cuda"
abl = int(sys.argv[1]) if len(sys.argv) > 1 else 0; use_sync = int(sys.argv[2]) if len(sys.argv) > 2 else 0
N, H, W, Cin = (int(v) for v in sys.argv[3:7]) if len(sys.argv) > 6 else (16, 192, 256, 67)
dll = ctypes.CDLL(_lib.LIB_PATH, mode=ctypes.RTLD_GLOBAL)
if abl:
    dll.cnm_tune_rows7s_ablate(abl)
x = ops.nchw_to_c4(torch.randn(N, Cin, H, W, device=dev)); wt = torch.randn(128, Cin, 7, 7, device=dev) * 0.02
up = ops.pack_winograd(wt, stride=1, tile=4); bp = torch.randn(128, device=dev)
sync = ops.wino36_sync_workspace(dev) if use_sync else None
fn = lambda: ops.conv_rows_winograd_c4(x, up, bp, 128, 7, True, stride=1, tile=4, sync=sync)
for _ in range(3): fn()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): fn()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
mf = 2.0 * N * H * W * 128 * (4 * ((Cin + 3) // 4)) * 7 * 10 / 4 / 1e9
mhz = 0.0
if hasattr(dll, "cnm_debug_rows7s_mhz"):
    dll.cnm_debug_rows7s_mhz.restype = ctypes.c_double; mhz = dll.cnm_debug_rows7s_mhz()
print("staged rows7 ablate %2d sync %d: %.3f ms  %.1f TF on the MFMAs (%.3f of 157.3)  shader clock %.0f MHz" % (abl, use_sync, ms, mf / ms, mf / ms / 157.3, mhz))

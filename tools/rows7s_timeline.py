"""Per-point timeline of one wave pair of the staged rows kernel (needs the -DROWS7S_TIMELINE build): cycles between the
starts of consecutive frequency points, phase end, barrier, for waves 0 and 4 of workgroup 0, eight phases in the middle."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from cnmnet_amd import ops, _lib
lib = _lib.load(); dev = "cuda"
dll = ctypes.CDLL(_lib.LIB_PATH, mode=ctypes.RTLD_GLOBAL)
N, H, W, Cin = 16, 192, 256, 67
x = ops.nchw_to_c4(torch.randn(N, Cin, H, W, device=dev)); wt = torch.randn(128, Cin, 7, 7, device=dev) * 0.02
up = ops.pack_winograd(wt, stride=1, tile=4); bp = torch.randn(128, device=dev)
fn = lambda: ops.conv_rows_winograd_c4(x, up, bp, 128, 7, True, stride=1, tile=4)
for _ in range(5): fn()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): fn()
e1.record(); torch.cuda.synchronize()
print("%.3f ms per launch with the timeline probes" % (e0.elapsed_time(e1) / 10))
buf = (ctypes.c_uint32 * (2 * 8 * 12))()
assert dll.cnm_debug_rows7s_timeline(buf) == 0
t = np.frombuffer(buf, dtype=np.uint32).reshape(2, 8, 12).astype(np.int64)
for w in range(2):
    print("wave %d: per phase [point 0 .. point 9 | wait for loads + barrier | to next phase start]; phase length" % (4 * w))
    for ph in range(8):
        d = np.diff(t[w, ph]) & 0xFFFFFFFF
        nxt = ((t[w, ph + 1, 0] - t[w, ph, 11]) & 0xFFFFFFFF) if ph + 1 < 8 else -1
        tot = ((t[w, ph + 1, 0] - t[w, ph, 0]) & 0xFFFFFFFF) if ph + 1 < 8 else -1
        print("  " + " ".join("%5d" % v for v in d[:10]) + " | %5d | %5d ; %6d" % (d[10], nxt, tot))
print("offset wave 4 - wave 0 at the phase starts:", [int((t[1, ph, 0] - t[0, ph, 0])) for ph in range(8)])

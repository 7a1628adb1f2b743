"""Per-step timeline of one wave pair of the staged 36-point kernel (needs the -DWINO4S_TIMELINE build):
python tools/wino36s_timeline.py Cin Cout H W N  -- cycles between the starts of consecutive double steps (18 per phase),
wait + barrier, for waves 0 and 4 of workgroup 0, eight phases in the middle of its range."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from cnmnet_amd import ops, _lib
Cin, Cout, H, W, N = [int(v) for v in sys.argv[1:6]] if len(sys.argv) > 5 else (256, 512, 48, 64, 16)
lib = _lib.load(); dll = ctypes.CDLL(_lib.LIB_PATH, mode=ctypes.RTLD_GLOBAL)
x = ops.nchw_to_c4(torch.randn(N, Cin, H, W, device="cuda")); up = ops.pack_winograd4(torch.randn(Cout, Cin, 3, 3, device="cuda") * 0.02); bp = torch.zeros(Cout, device="cuda")
fn = lambda: ops.conv3x3_winograd4_c4(x, up, bp, Cout, True)
for _ in range(5): fn()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): fn()
e1.record(); torch.cuda.synchronize()
print("%d->%d %dx%d N%d: %.4f ms per launch with the timeline probes" % (Cin, Cout, H, W, N, e0.elapsed_time(e1) / 10))
buf = (ctypes.c_uint32 * (2 * 8 * 24))()
assert dll.cnm_debug_wino4s_timeline(buf) == 0
t = np.frombuffer(buf, dtype=np.uint32).reshape(2, 8, 24).astype(np.int64)
for w in range(2):
    print("wave %d: per phase [double step 0 .. 17 | wait for loads + barrier | to next phase start]; phase length" % (4 * w))
    for ph in range(8):
        d = np.diff(t[w, ph, :20]) & 0xFFFFFFFF
        k = 0 if w == 0 else 6                                  # the wave's DMA step
        dm = [(t[w, ph, 20] - t[w, ph, k]) & 0xFFFFFFFF, (t[w, ph, 21] - t[w, ph, 20]) & 0xFFFFFFFF, (t[w, ph, 22] - t[w, ph, 21]) & 0xFFFFFFFF, (t[w, ph, k + 1] - t[w, ph, 22]) & 0xFFFFFFFF]
        nxt = ((t[w, ph + 1, 0] - t[w, ph, 19]) & 0xFFFFFFFF) if ph + 1 < 8 else -1
        tot = ((t[w, ph + 1, 0] - t[w, ph, 0]) & 0xFFFFFFFF) if ph + 1 < 8 else -1
        print("  " + " ".join("%4d" % v for v in d[:18]) + " | %5d | %5d ; %6d" % (d[18], nxt, tot) + "   DMA step: before %d, pieces %d, advance %d, rest %d" % tuple(dm))
print("offset wave 4 - wave 0 at the phase starts:", [int((t[1, ph, 0] - t[0, ph, 0])) for ph in range(8)])

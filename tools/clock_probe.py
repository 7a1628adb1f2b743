"""Shader clock and power while one kernel family runs back to back (rocm-smi sampled from a child process):
python tools/clock_probe.py f16|f32|f32s|sweep"""
import os, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cnmnet_amd import ops
kind = sys.argv[1] if len(sys.argv) > 1 else "f16"
dev = "cuda"
if kind == "f16":
    x = ops.nchw_to_c8(torch.randn(16, 128, 96, 128, device=dev)); wp, bp = ops.pack_conv_f16(torch.randn(256, 128, 5, 5, device=dev) * 0.02, None, torch.zeros(256, device=dev))
    fn = lambda: ops.conv2d_c8(x, wp, bp, 256, 5, 1, True)
elif kind == "f32":
    x = ops.nchw_to_c4(torch.randn(16, 256, 48, 64, device=dev)); up = ops.pack_winograd4(torch.randn(512, 256, 3, 3, device=dev) * 0.02); bp = torch.zeros(512, device=dev)
    fn = lambda: ops.conv3x3_winograd4_c4(x, up, bp, 512, True)
elif kind == "f32s":                                                     # the staged persistent kernel (round 3+), stream-K ranges
    x = ops.nchw_to_c4(torch.randn(16, 256, 48, 64, device=dev)); up = ops.pack_winograd4(torch.randn(512, 256, 3, 3, device=dev) * 0.02); bp = torch.zeros(512, device=dev)
    sync = ops.wino36_sync_workspace(dev)
    fn = lambda: ops.conv3x3_winograd4_c4(x, up, bp, 512, True, sync=sync)
else:
    from cnmnet_amd import synthetic as syn
    img, cams = syn.frames(8, 2, 192, 256); img, cams = torch.from_numpy(img).to(dev), torch.from_numpy(cams).to(dev)
    hmkt = ops.homography_terms(cams[:, 0], cams[:, 1:])
    fn = lambda: ops.plane_sweep_cat_c4(img[:, 0], img[:, 1:], hmkt, 3.0, 64)
samples = []
def sample():
    for _ in range(6):
        time.sleep(0.5)
        try:
            o = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=10).stdout
            samples.append(" | ".join(l.strip() for l in o.splitlines() if "sclk" in l or "Power" in l or "mclk" in l))
        except Exception as e:
            samples.append(str(e))
th = threading.Thread(target=sample); th.start()
t0 = time.time(); n = 0
while th.is_alive():
    for _ in range(50): fn()
    torch.cuda.synchronize(); n += 50
dt = time.time() - t0
print(kind, "%d launches, %.1f us each" % (n, dt / n * 1e6))
for s in samples: print("  ", s[:300])

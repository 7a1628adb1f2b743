"""Frame-pipeline time (bench workload: 8 frames, 1 ref + 2 src, 192x256, 64 planes) against the executors'
F(4x4,3x3) / F(2x2,3x3) switch point (cnm_tune_wino4_min_workgroups)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cnmnet_amd import _lib, synthetic as syn
from cnmnet_amd.depthnet import depthNet, DepthRefineNet
from cnmnet_amd.pipeline import FramePipeline
dev = torch.device("cuda:0")
img, cams = syn.frames(8, 2, 192, 256, seed=7)
img, cams = torch.from_numpy(img).to(dev), torch.from_numpy(cams).to(dev)
pipe = FramePipeline(depthNet(3.0, 64).to(dev).eval(), DepthRefineNet(32, 3.0).to(dev).eval(), k_size=9)
lib = _lib.load()
for thr in (96, 192, 256, 384, 512, 768, 1536, 1 << 30):
    lib.cnm_tune_wino4_min_workgroups(thr)
    for _ in range(3): pipe(img, cams)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(10): pipe(img, cams)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 10
    print("min workgroups %10d: %.2f ms/step -> %.1f frames/s" % (thr, dt * 1e3, 8 / dt))

"""BASELINE config 4: 640x480, 96 planes, 1 ref + 4 src, batch 4 on one MI355X (HBM-bound stress).
Runs the frame pipeline, checks a size-independent property (two identical sources on the same side give the
same refine input as S=2 with those sources) and reports timing."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cnmnet_amd import synthetic as syn
from cnmnet_amd.depthnet import depthNet, DepthRefineNet
from cnmnet_amd.pipeline import FramePipeline
dev = torch.device("cuda:0")
B, S, H, W, D = 4, 4, 480, 640, 96
img, cams = syn.frames(B, S, H, W, seed=7)
pipe = FramePipeline(depthNet(3.0, D).to(dev).eval(), DepthRefineNet(32, 3.0).to(dev).eval(), k_size=9)
img, cams = torch.from_numpy(img).to(dev), torch.from_numpy(cams).to(dev)
out = pipe(img, cams); torch.cuda.synchronize()
t = time.perf_counter(); n = 3
for _ in range(n): out = pipe(img, cams)
torch.cuda.synchronize(); dt = (time.perf_counter() - t) / n
print("config 4: %.1f ms per batch of %d frames (S=%d, %dx%d, D=%d) -> %.2f frames/s; peak mem %.1f GB; finite %s" % (
    dt * 1e3, B, S, W, H, D, B / dt, torch.cuda.max_memory_allocated() / 2**30, bool(torch.isfinite(out["disp"]).all() and torch.isfinite(out["normal"]).all())))
# property: duplicating the sources (s0,s1,s0,s1) must reproduce the S=2 result of (s0,s1): averages of equal things
img2 = torch.cat((img[:1, :3], img[:1, 1:3]), 1); cams2 = torch.cat((cams[:1, :3], cams[:1, 1:3]), 1)
a = pipe(img2, cams2); b = pipe(img[:1, :3].contiguous(), cams[:1, :3].contiguous())
print("duplicate-source property: max |disp(S=4 dup) - disp(S=2)| = %.2e" % float((a["disp"] - b["disp"]).abs().max()))

"""Probe: frame pipeline with the fused up_conv layers on / off per net (GPU box only)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from cnmnet_amd import synthetic as syn
from cnmnet_amd.depthnet import depthNet, DepthRefineNet
from cnmnet_amd.pipeline import FramePipeline
dev = torch.device("cuda:0")
dn = bench.load_weights(depthNet(3.0, bench.PLANES), 1).to(dev)
rn = bench.load_weights(DepthRefineNet(32, 3.0), 2).to(dev)
pipe = FramePipeline(dn, rn, k_size=9, normals=True)
img, cams = syn.frames(8, 2, bench.H, bench.W, seed=1)
img, cams = torch.from_numpy(img).to(dev), torch.from_numpy(cams).to(dev)
from cnmnet_amd import _lib
for rep, thr in enumerate((196608, 98304, 196608, 98304)):
    _lib.load().cnm_tune_upsampled_min_pixels(thr); print('threshold', thr)
    for fd, fr in ((True, True),):
        dn.fused_upsample, rn.fused_upsample = fd, fr
        for _ in range(3): pipe(img, cams)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(20): pipe(img, cams)
        torch.cuda.synchronize()
        print("fused depthNet %-5s refine %-5s: %.3f ms/step" % (fd, fr, (time.perf_counter() - t) / 20 * 1e3))

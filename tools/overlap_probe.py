"""Probe: depthNet of batch k+1 on one stream while DepthRefineNet + normals of batch k run on another (GPU box only)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from cnmnet_amd import synthetic as syn, ops
from cnmnet_amd.depthnet import depthNet, DepthRefineNet
from cnmnet_amd.pipeline import FramePipeline

dev = torch.device("cuda:0")
B, S, H, W = 8, 2, bench.H, bench.W
dn = bench.load_weights(depthNet(3.0, bench.PLANES), 1).to(dev)
rn = bench.load_weights(DepthRefineNet(32, 3.0), 2).to(dev)
pipe = FramePipeline(dn, rn, k_size=9, normals=True)
img, cams = syn.frames(B, S, H, W, seed=1)
img, cams = torch.from_numpy(img).to(dev), torch.from_numpy(cams).to(dev)
K = 20
for _ in range(3): pipe(img, cams)
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(K): pipe(img, cams)
torch.cuda.synchronize(); serial = (time.perf_counter() - t) / K

sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
HW = H * W
def stage_a():
    return dn.forward_pairs(img[:, 0], img[:, 1:], cams[:, 0], cams[:, 1:])
def stage_b(dp, feat):
    flat = dp[0].view(-1)
    disp, prob, _ = rn.forward_c4(flat, flat[HW:], S * HW, feat, S * 16, 0, feat, S * 16, 16, B, H, W)
    k_inv = ops.intrinsics_inverse(cams[:, 0])
    return ops.depth2normal(disp.view(B, H, W), k_inv, 9, input_is_idepth=True)
keep = []
def run(n):
    prev = None
    for k in range(n + 1):
        ev = None
        if k < n:
            with torch.cuda.stream(sA):
                cur = stage_a(); ev = torch.cuda.Event(); ev.record(sA)
        if prev is not None:
            with torch.cuda.stream(sB):
                sB.wait_event(prev[1]); keep.append(stage_b(*prev[0]))
        prev = (cur, ev) if k < n else None
        keep.append(cur)
run(3); torch.cuda.synchronize(); keep.clear()
t = time.perf_counter(); run(K); torch.cuda.synchronize(); over = (time.perf_counter() - t) / K
print("serial %.3f ms/step  overlapped %.3f ms/step  (%.1f%%)" % (serial * 1e3, over * 1e3, 100 * (serial / over - 1)))

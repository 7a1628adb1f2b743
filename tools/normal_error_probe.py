"""Probe: error of the frame outputs against the CPU oracle at the bench configuration for several kernel selections."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from cnmnet_amd import synthetic as syn, _lib, ops
from cnmnet_amd.depthnet import depthNet, DepthRefineNet
from cnmnet_amd.pipeline import FramePipeline
from oracle import ref_arrangement as ra
from conftest import torch_state
T = torch.from_numpy
dev = torch.device("cuda:0")
def load(m, seed):
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    hs = 0.2 if seed == 41 else 0.05                                     # heads out of saturation at this size (tests/test_gpu_parity.py)
    w = syn.state_dict_like(shapes, seed=seed, randomize_bn=True)
    w = {k: (v * hs if (v.ndim == 4 and v.shape[0] == 1) else v) for k, v in w.items()}
    m.load_state_dict(torch_state(w)); return m.eval()
B, S, H, W = 8, 2, 192, 256
img, cams = syn.frames(B, S, H, W, seed=1234)
b = 0
with torch.no_grad():
    want = ra.frame_forward(load(ra.DepthNetCPU(3.0), 41), load(ra.DepthRefineNetCPU(32, 3.0), 42), T(img[b:b+1, 0]), T(img[b:b+1, 1]), T(img[b:b+1, 2]),
                            T(cams[b:b+1, 0]), T(cams[b:b+1, 1]), T(cams[b:b+1, 2]))
def stats(e):
    e = e.flatten().double()
    return "med %.1e q90 %.1e q99 %.1e q999 %.1e max %.1e" % tuple(float(e.quantile(q)) if q < 1 else float(e.max()) for q in (0.5, 0.9, 0.99, 0.999, 1))
for name, wino, wino4, fused in (("default", True, True, True), ("no fused upsample", True, True, False), ("F(2x2)+rows only", True, False, False), ("direct kernels", False, False, False)):
    dn, rn = load(depthNet(3.0), 41).to(dev), load(DepthRefineNet(32, 3.0), 42).to(dev)
    for n in (dn, rn): n.winograd, n.winograd4, n.fused_upsample = wino, wino4, fused
    with torch.no_grad():
        out = FramePipeline(dn, rn, k_size=9)(T(img).to(dev), T(cams).to(dev))
    print("%-20s disp: %s" % (name, stats((out["disp"][b:b+1].cpu() - want["disp"]).abs())))
    print("%-20s nrm : %s" % (name, stats((out["normal"][b:b+1].cpu() - want["normal"]).abs().amax(1))))
# normals from the ORACLE's inverse depth through the GPU kernel: the depth->normal kernel alone
k_inv = ops.intrinsics_inverse(T(cams[b:b+1, 0]).to(dev))
n_gpu, _ = ops.depth2normal(want["disp"].view(1, H, W).to(dev), k_inv, 9, input_is_idepth=True)
print("%-20s nrm : %s" % ("K6 on oracle depth", stats((n_gpu.cpu() - want["normal"]).abs().amax(1))))
d = want["disp"].flatten()
print("oracle disp quantiles:", [round(float(d.quantile(q)), 4) for q in (0, 0.01, 0.1, 0.5, 0.9, 0.99, 1)], " disp_a:", [round(float(want["disp_a"].flatten().quantile(q)), 4) for q in (0, 0.5, 1)])
print("prob quantiles:", [round(float(want["prob"].flatten().quantile(q)), 4) for q in (0, 0.5, 1)], " normals nonzero fraction %.3f" % float((want["normal"].abs().amax(1) > 0).float().mean()))
# the same depth -> normal fit evaluated in float64 (the reference's arrangement, double arithmetic): whose rounding is it?
depth64 = 1.0 / want["disp"].double().squeeze(1)
n64, _ = ra.depth_to_normal(depth64, T(cams[b:b+1, 0])[:, 1, :3, :3].double().inverse(), 9)
print("%-20s nrm : %s" % ("fp32 oracle vs fp64", stats((want["normal"].double() - n64).abs().amax(1))))
print("%-20s nrm : %s" % ("GPU K6 vs fp64", stats((n_gpu.cpu().double() - n64).abs().amax(1))))

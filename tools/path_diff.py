"""|default (Winograd) path - all-direct fp32 path| on the refined inverse depth, per weight initialisation and size."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from cnmnet_amd import synthetic as syn
from cnmnet_amd.depthnet import depthNet, DepthRefineNet
from cnmnet_amd.pipeline import FramePipeline
from conftest import torch_state
dev = torch.device("cuda:0")
def load(m, seed):
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    m.load_state_dict(torch_state(syn.state_dict_like(shapes, seed=seed, randomize_bn=True))); return m.eval()
for (B, S, H, W, D) in ((2, 2, 192, 256, 64), (1, 4, 480, 640, 96), (1, 2, 480, 640, 96)):
    img, cams = syn.frames(B, S, H, W, seed=7)
    img, cams = torch.from_numpy(img).to(dev), torch.from_numpy(cams).to(dev)
    for init in ("default-init", "seeded"):
        torch.manual_seed(0)
        dn, rn = depthNet(3.0, D).to(dev).eval(), DepthRefineNet(32, 3.0).to(dev).eval()
        if init == "seeded": dn, rn = load(dn, 3).to(dev), load(rn, 4).to(dev)
        outs = {}
        for tag, w, w4 in (("direct", False, False), ("F(2x2)/F(2,k)", True, False), ("default", True, True)):
            dn.winograd = rn.winograd = w; dn.winograd4 = rn.winograd4 = w4
            outs[tag] = FramePipeline(dn, rn, k_size=9)(img, cams)["disp"].float().cpu()
        d0 = outs["direct"]
        print("%dx%d S=%d D=%d %-12s disp range [%.3f, %.3f]: |F2 - direct| %.2e, |default - direct| %.2e" % (
            W, H, S, D, init, float(d0.min()), float(d0.max()), float((outs["F(2x2)/F(2,k)"] - d0).abs().max()), float((outs["default"] - d0).abs().max())))

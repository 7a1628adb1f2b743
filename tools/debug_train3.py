import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch, torch.nn.functional as F
from cnmnet_amd import synthetic as syn, ops, autograd as ag
from cnmnet_amd.depthnet import depthNet
from conftest import torch_state
T = torch.from_numpy; dev = torch.device("cuda:0")
def load(m, seed):
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    m.load_state_dict(torch_state(syn.state_dict_like(shapes, seed=seed, randomize_bn=True))); return m
img, cams = syn.frames(2, 1, 64, 64, seed=404)
calls = []
orig_bwd = ag.BatchNormReLUC4.backward
def bwd(ctx, dy):
    x, y, gamma, mean, invstd = ctx.saved_tensors
    out = orig_bwd(ctx, dy)
    C = gamma.numel()
    calls.append(dict(x=ops.c4_to_nchw(x, C).cpu().double(), dy=ops.c4_to_nchw(dy.contiguous(), C).cpu().double(), g=gamma.detach().cpu().double(),
                      dx=ops.c4_to_nchw(out[0], C).cpu().double(), dg=out[1].cpu().double(), db=out[2].cpu().double(), y=ops.c4_to_nchw(y, C).cpu().double()))
    return out
ag.BatchNormReLUC4.backward = staticmethod(bwd)
gn = load(depthNet(3.0), 61).to(dev).train()
o2, f2 = gn(*(T(a).to(dev) for a in (img[:, 0], img[:, 1], cams[:, 0], cams[:, 1])))
o2[3].mean().backward()
for i, c in enumerate(calls):
    x, dy, g = c["x"], c["dy"], c["g"]
    m = x.shape[0] * x.shape[2] * x.shape[3]
    mu = x.mean((0, 2, 3), keepdim=True); var = x.var((0, 2, 3), unbiased=False, keepdim=True)
    istd = 1.0 / torch.sqrt(var + 1e-5); xh = (x - mu) * istd
    d = dy * (c["y"] > 0)
    sd = d.sum((0, 2, 3), keepdim=True); sq = (d * xh).sum((0, 2, 3), keepdim=True)
    dx = g.view(1, -1, 1, 1) * istd * (d - sd / m - xh * sq / m)
    r = lambda a, bb: float((a - bb).abs().max() / (bb.abs().max() + 1e-30))
    print("bn bwd call %2d shape %-18s dx %.1e dgamma %.1e dbeta %.1e |dy|max %.1e min istd^-1 %.1e" % (i, tuple(x.shape), r(c["dx"], dx), r(c["dg"], sq.flatten()), r(c["db"], sd.flatten()), float(dy.abs().max()), float(1 / istd.max())))

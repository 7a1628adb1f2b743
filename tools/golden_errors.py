"""Max |error| of the engine's default fp32 path against the reference's golden outputs (tests/golden/*.npz), per output."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from cnmnet_amd import synthetic as syn, _lib
from cnmnet_amd.depthnet import depthNet, DepthRefineNet
from conftest import torch_state
dev = torch.device("cuda:0"); T = torch.from_numpy
def load(m, seed):
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    m.load_state_dict(torch_state(syn.state_dict_like(shapes, seed=seed, randomize_bn=True))); return m.eval()
g, gr = np.load(os.path.join(ROOT, "tests/golden/depthnet_64x96.npz")), np.load(os.path.join(ROOT, "tests/golden/refine_64x96.npz"))
img, cams = syn.frames(2, 2, 64, 96, seed=int(g["seed"]))
for thr, tag in ((384, "default dispatch"), (1, "every eligible layer on the 36-point kernels")):
    _lib.load().cnm_tune_wino4_min_workgroups(thr)
    net = load(depthNet(3.0), int(g["weight_seed"])).to(dev); ref = load(DepthRefineNet(32, 3.0), int(gr["weight_seed"])).to(dev)
    L, lc = T(img[:, 0]).to(dev), T(cams[:, 0]).to(dev)
    with torch.no_grad():
        o, f = net(L, T(img[:, 1]).to(dev), lc, T(cams[:, 1]).to(dev)); ob, fb = net(L, T(img[:, 2]).to(dev), lc, T(cams[:, 2]).to(dev))
        d, p, vf = ref(idepth01=o[0], idepth02=ob[0], iconv01=f, iconv02=fb, ReturnVolume=True)
    ch = list(g["iconv1_channels"])
    e = {"disp%d" % (i + 1): float(np.abs(o[i].cpu().numpy() - g["disp%d" % (i + 1)]).max()) for i in range(4)}
    e["iconv1/scale"] = float(np.abs(f[:, ch].cpu().numpy() - g["iconv1"]).max() / np.abs(g["iconv1"]).max())
    e["refined"] = float(np.abs(d.cpu().numpy() - gr["disp_refined"]).max()); e["prob"] = float(np.abs(p.cpu().numpy() - gr["prob_map"]).max())
    e["iconv1_depth/scale"] = float(np.abs(vf[:, ch].cpu().numpy() - gr["iconv1_depth"]).max() / np.abs(gr["iconv1_depth"]).max())
    print(tag + ":", {k: "%.1e" % v for k, v in e.items()})

import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from cnmnet_amd import synthetic as syn
from cnmnet_amd.depthnet import depthNet, DepthRefineNet
from conftest import torch_state
T = torch.from_numpy; dev = torch.device("cuda:0")
G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
g, gr = dict(np.load(G + "/depthnet_64x96.npz")), dict(np.load(G + "/refine_64x96.npz"))
def load(m, seed):
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    m.load_state_dict(torch_state(syn.state_dict_like(shapes, seed=seed, randomize_bn=True))); return m.eval()
img, cams = syn.frames(2, 2, 64, 96, seed=int(g["seed"]))
for prec in ("f32", "f16"):
    dn = load(depthNet(3.0, 64, precision=prec), int(g["weight_seed"])).to(dev); rn = load(DepthRefineNet(32, 3.0, precision=prec), int(gr["weight_seed"])).to(dev)
    L, lc = T(img[:, 0]).to(dev), T(cams[:, 0]).to(dev)
    with torch.no_grad():
        o1, f1 = dn(L, T(img[:, 1]).to(dev), lc, T(cams[:, 1]).to(dev)); o2, f2 = dn(L, T(img[:, 2]).to(dev), lc, T(cams[:, 2]).to(dev))
        disp, prob = rn(idepth01=o1[0], idepth02=o2[0], iconv01=f1, iconv02=f2)
    e = lambda a, b: (float(np.abs(a.cpu().numpy() - b).max()), float(np.quantile(np.abs(a.cpu().numpy() - b), 0.999)))
    ch = list(g["iconv1_channels"])
    print(prec, "disp1..4", [e(o1[i], g["disp%d" % (i + 1)]) for i in range(4)], "iconv rel", e(f1[:, ch], g["iconv1"])[0] / np.abs(g["iconv1"]).max(),
          "refined", e(disp, gr["disp_refined"]), "prob", e(prob, gr["prob_map"]))

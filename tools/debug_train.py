import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from cnmnet_amd import synthetic as syn
from cnmnet_amd.depthnet import depthNet, DepthRefineNet
from oracle import ref_arrangement as ra
from conftest import torch_state
T = torch.from_numpy
dev = torch.device("cuda:0")
def load(m, seed):
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    m.load_state_dict(torch_state(syn.state_dict_like(shapes, seed=seed, randomize_bn=True))); return m
img, cams = syn.frames(2, 1, 64, 64, seed=404)
res = []
mode = sys.argv[1] if len(sys.argv) > 1 else "d1"
for make, device, dt in ((lambda: ra.DepthNetCPU(3.0), torch.device("cpu"), torch.float64), (lambda: ra.DepthNetCPU(3.0), torch.device("cpu"), torch.float32), (lambda: depthNet(3.0), dev, torch.float32)):
    dn = load(make(), 61).to(device).to(dt).train()
    o, f = dn(T(img[:, 0]).to(device).to(dt), T(img[:, 1]).to(device).to(dt), T(cams[:, 0]).to(device).to(dt), T(cams[:, 1]).to(device).to(dt))
    loss = {"d1": lambda: o[0].mean(), "d4": lambda: o[3].mean(), "f": lambda: f.mean(), "d2": lambda: o[1].mean()}[mode]()
    loss.backward()
    res.append({k: p.grad.detach().cpu().numpy() for k, p in dn.named_parameters() if p.grad is not None})
for k in res[0]:
    b = res[0][k]
    e = [np.abs(r[k] - b).max() / (np.abs(b).max() + 1e-30) for r in res[1:]]
    print("%-22s cpu32-vs-f64 %.2e   gpu-vs-f64 %.2e" % (k, e[0], e[1]))

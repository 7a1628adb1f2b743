import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from cnmnet_amd import synthetic as syn, ops, autograd as ag
from cnmnet_amd.depthnet import depthNet
from oracle import ref_arrangement as ra
from conftest import torch_state
T = torch.from_numpy; dev = torch.device("cuda:0")
def load(m, seed):
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    m.load_state_dict(torch_state(syn.state_dict_like(shapes, seed=seed, randomize_bn=True))); return m
img, cams = syn.frames(2, 1, 64, 64, seed=404)
# oracle fp64 with hooks on every ReLU output
cpu_acts, cpu_grads = [], {}
dn = load(ra.DepthNetCPU(3.0), 61).double().train()
def mk(i):
    def hook(mod, inp, out):
        idx = len(cpu_acts); cpu_acts.append(out.detach().numpy().copy())
        out.register_hook(lambda g, idx=idx: cpu_grads.__setitem__(idx, g.detach().numpy().copy()))
    return hook
for m in dn.modules():
    if isinstance(m, torch.nn.ReLU): m.register_forward_hook(mk(0))
o, f = dn(*(T(a).double() for a in (img[:, 0], img[:, 1], cams[:, 0], cams[:, 1])))
o[3].mean().backward()
# gpu with patched conv_bn_relu
gpu_acts, gpu_grads = [], {}
orig = ag.conv_bn_relu
def patched(x, conv, bn, rot=0):
    out = orig(x, conv, bn, rot)
    idx = len(gpu_acts); gpu_acts.append((ops.c4_to_nchw(out.detach(), conv.out_channels).cpu().numpy()))
    out.register_hook(lambda g, idx=idx, c=conv.out_channels: gpu_grads.__setitem__(idx, ops.c4_to_nchw(g.contiguous(), c).cpu().numpy()))
    return out
ag.conv_bn_relu = patched
gn = load(depthNet(3.0), 61).to(dev).train()
o2, f2 = gn(*(T(a).to(dev) for a in (img[:, 0], img[:, 1], cams[:, 0], cams[:, 1])))
o2[3].mean().backward()
names = ["conv1.0","conv1.3","conv2.0","conv2.3","conv3.0","conv3.3","conv4.0","conv4.3","conv5.0","conv5.3","upconv5","iconv5","upconv4","iconv4","upconv3","iconv3","upconv2","iconv2","upconv1","iconv1"]
for i, n in enumerate(names):
    a, b = gpu_acts[i], cpu_acts[i]
    line = "%-8s act rel %.1e" % (n, np.abs(a - b).max() / (np.abs(b).max() + 1e-30))
    if i in cpu_grads and i in gpu_grads:
        ga, gb = gpu_grads[i], cpu_grads[i]
        line += "  grad rel %.1e (|g|max %.1e)" % (np.abs(ga - gb).max() / (np.abs(gb).max() + 1e-30), np.abs(gb).max())
    print(line, a.shape)

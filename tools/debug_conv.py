import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.nn.functional as F
from cnmnet_amd import ops, autograd as ag
T = torch.from_numpy; dev = torch.device("cuda:0")
def rel(a, b): return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))
for (cin, cout, k, stride, N, H, W) in [(1024, 512, 3, 1, 2, 8, 8), (512, 512, 3, 1, 2, 8, 8), (512, 512, 3, 2, 2, 4, 4), (512, 512, 3, 1, 2, 4, 4),
                                         (512, 512, 3, 2, 2, 8, 8), (256, 512, 3, 1, 2, 16, 16), (1024, 512, 3, 1, 2, 4, 4), (512, 256, 3, 1, 2, 16, 16)]:
    rng = np.random.default_rng(1)
    x = T(rng.standard_normal((N, cin, H, W)).astype(np.float32)).requires_grad_(True)
    w = T((rng.standard_normal((cout, cin, k, k)) * 0.05).astype(np.float32)).requires_grad_(True)
    y = F.conv2d(x, w, stride=stride, padding=(k - 1) // 2)
    gy = T(rng.standard_normal(tuple(y.shape)).astype(np.float32)); y.backward(gy)
    xd = ops.nchw_to_c4(x.detach().to(dev)).requires_grad_(True); wd = w.detach().to(dev).requires_grad_(True)
    yd = ag.ConvC4.apply(xd, wd, stride, 0); yd.backward(ops.nchw_to_c4(gy.to(dev)))
    print((cin, cout, k, stride, N, H, W), "fwd %.1e dgrad %.1e wgrad %.1e" % (rel(ops.c4_to_nchw(yd.detach(), cout).cpu().numpy(), y.detach().numpy()),
          rel(ops.c4_to_nchw(xd.grad, cin).cpu().numpy(), x.grad.numpy()), rel(wd.grad.cpu().numpy(), w.grad.numpy())))

"""GPU box: plane-sweep golden pairs -- engine vs the reference's fp32 output and vs the float64 closed form."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cnmnet_amd import ops
from oracle import closed_form as cf
g = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "planesweep_32x64.npz"))
dev = torch.device("cuda:0")
T = lambda a: torch.from_numpy(a).to(dev)
vol = ops.plane_sweep_volume(T(g["left"]), T(g["right"]), T(g["left_cam"]), T(g["right_cam"]), 3.0, 64).cpu().numpy()
ex = cf.plane_sweep_volume(g["left"], g["right"], g["left_cam"], g["right_cam"], 3.0, 64)
st = lambda e: "median %.2e q99.9 %.2e max %.2e" % (np.median(e), np.quantile(e, 0.999), e.max())
for p in range(2):
    print("pair %d: engine vs reference fp32: %s | engine vs float64: %s | reference vs float64: %s" % (
        p, st(np.abs(vol[p] - g["volume"][p])), st(np.abs(vol[p] - ex[p])), st(np.abs(g["volume"][p] - ex[p]))))
    e = np.abs(vol[p] - ex[p]); i = np.unravel_index(np.argmax(e), e.shape)
    print("   worst at plane %d pixel (%d,%d): engine %.6f float64 %.6f reference %.6f" % (i[0], i[1], i[2], vol[p][i], ex[p][i], g["volume"][p][i]))

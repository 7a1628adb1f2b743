"""Row-wise Winograd kernels: 128-cout workgroups (8 waves, cnm_tune_rows_wide) against 64-cout workgroups -- bit equality and
time at the bench shapes.  GPU box only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cnmnet_amd import ops, _lib
lib = _lib.load(); dev = "cuda"
CASES = [("depth conv1.0 7x7 s1", 16, 67, 128, 192, 256, 7, 1), ("depth conv1.3 7x7 s2", 16, 128, 128, 192, 256, 7, 2), ("depth conv2.3 5x5 s2", 16, 256, 256, 96, 128, 5, 2),
         ("depth conv3.3 3x3 s2", 16, 512, 512, 48, 64, 3, 2), ("refine conv1.3 3x3 s2", 8, 128, 128, 192, 256, 3, 2), ("refine conv2.3 3x3 s2", 8, 256, 256, 96, 128, 3, 2),
         ("ragged 7x7 s1", 2, 35, 128, 37, 50, 7, 1), ("ragged 5x5 s2", 3, 64, 256, 30, 44, 5, 2)]
def ev(fn, iters=20, warm=3):
    for _ in range(warm): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
bad = 0; tot = [0.0, 0.0]
for name, N, Cin, Cout, H, W, k, st in CASES:
    x = ops.nchw_to_c4(torch.randn(N, Cin, H, W, device=dev)); wt = torch.randn(Cout, Cin, k, k, device=dev) * 0.02
    up = ops.pack_winograd_rows(wt, stride=2, tile=4) if k == 3 else ops.pack_winograd(wt, stride=st, tile=4)
    bp = torch.randn(Cout, device=dev)
    fn = lambda: ops.conv_rows_winograd_c4(x, up, bp, Cout, k, True, stride=st, tile=4)
    outs, ms = [], []
    for mode in (0, 2):
        lib.cnm_tune_rows_wide(mode); outs.append(fn().clone())
    for rnd in range(2):
        for mode in (0, 2):
            lib.cnm_tune_rows_wide(mode); m = ev(fn)
            if rnd: ms.append(m)
    lib.cnm_tune_rows_wide(1)
    eq = torch.equal(outs[0], outs[1]); bad += not eq
    if not name.startswith("ragged"): tot[0] += ms[0]; tot[1] += ms[1]
    print("%-24s N%2d %4d->%4d %3dx%-3d: equal %s | 64-cout %.3f ms | 128-cout %.3f ms  x%.2f" % (name, N, Cin, Cout, H, W, eq, ms[0], ms[1], ms[0] / ms[1]), flush=True)
print("sum (bench shapes): %.3f -> %.3f ms   %s" % (tot[0], tot[1], "CHECK FAILED" if bad else "CHECK OK"))

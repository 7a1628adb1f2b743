"""LDS-staged 7x7 stride-1 row-wise Winograd kernel (conv_rows_staged.hip) against the gather-fed one: bit equality without a
sync workspace, closeness + run-to-run reproducibility with one, and time.  GPU box only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cnmnet_amd import ops, _lib
lib = _lib.load(); dev = "cuda"
CASES = [("ragged small", 2, 35, 128, 37, 50, None), ("one unit", 1, 8, 128, 4, 32, None), ("odd chunks", 1, 12, 128, 9, 40, None), ("two inputs", 2, 64, 256, 21, 70, 3),
         ("depth conv1.0", 16, 67, 128, 192, 256, None), ("depth conv1.0 cfg4", 4, 67, 128, 480, 640, None)]
def ev(fn, iters=20, warm=3):
    for _ in range(warm): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
bad = 0
sync = ops.wino36_sync_workspace(dev)
for name, N, Cin, Cout, H, W, split in CASES:
    torch.manual_seed(1)
    xs = torch.randn(N, Cin, H, W, device=dev)
    if split:
        x = ops.nchw_to_c4(xs[:, :4 * split]); x2 = ops.nchw_to_c4(xs[:, 4 * split:])
    else:
        x = ops.nchw_to_c4(xs); x2 = None
    wt = torch.randn(Cout, Cin, 7, 7, device=dev) * 0.02
    up = ops.pack_winograd(wt, stride=1, tile=4); bp = torch.randn(Cout, device=dev)
    fn = lambda s=None: ops.conv_rows_winograd_c4(x, up, bp, Cout, 7, True, x2=x2, stride=1, tile=4, sync=s)
    lib.cnm_tune_rows7_staged(0); ref = fn().clone(); t0 = ev(fn)
    lib.cnm_tune_rows7_staged(1); a = fn().clone(); t1 = ev(fn)
    b = fn(sync).clone(); b2 = fn(sync).clone(); t2 = ev(lambda: fn(sync))
    torch.cuda.synchronize()
    eq = torch.equal(ref, a); close = torch.allclose(ref, b, rtol=1e-4, atol=1e-4); rep = torch.equal(b, b2); clean = ops.sync_workspace_state(sync)[1] == 0
    ok = eq and close and rep and clean; bad += not ok
    fl = 2.0 * N * H * W * Cout * Cin * 49 / 1e9
    print("%-20s N%2d %3d->%3d %3dx%-3d: equal %s sync-close %s (max %.2e) repro %s flags-zero %s | gather %.3f ms | staged %.3f ms (x%.2f) | +sync %.3f ms (x%.2f, %.1f TF direct-equivalent)"
          % (name, N, Cin, Cout, H, W, eq, close, (ref - b).abs().max().item(), rep, clean, t0, t1, t0 / t1, t2, t0 / t2, fl / t2), flush=True)
print("CHECK FAILED" if bad else "CHECK OK")

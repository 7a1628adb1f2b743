"""Split path of the staged rows kernel: head part and published part checked separately against unsplit runs with the
other part's filter chunks zeroed (needs the -DROWS7S_ABLATE build).  GPU box only."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cnmnet_amd import ops, _lib
lib = _lib.load(); dev = "cuda"
abl = ctypes.CDLL(_lib.LIB_PATH, mode=ctypes.RTLD_GLOBAL).cnm_tune_rows7s_ablate
sync = ops.wino36_sync_workspace(dev)
N, Cin, H, W = 1, 32, 4, 64                                   # 2 units x 7 phases, grid 3: begins 0, 4, 9, 14
torch.manual_seed(1)
x = ops.nchw_to_c4(torch.randn(N, Cin, H, W, device=dev)); wt = torch.randn(128, Cin, 7, 7, device=dev) * 0.02
up = ops.pack_winograd(wt, stride=1, tile=4); bp = torch.randn(128, device=dev)
run = lambda u, s=None: ops.conv_rows_winograd_c4(x, u, bp, 128, 7, False, stride=1, tile=4, sync=s)
upc = up.view(-1, 8, 10, 64, 4)                               # [chunk16][cout/16][xi][lane][4]
def masked(lo, hi):
    m = torch.zeros_like(upc); m[lo:hi] = upc[lo:hi]; return m.reshape(-1)
for unit, split in ((0, 8), (1, 4)):
    cols = slice(32 * unit, 32 * unit + 32)
    head_ref = run(masked(0, split))[..., cols, :]; tail_ref = run(masked(split, 14))[..., cols, :] - bp.view(1, 32, 1, 1, 4)
    abl(64); head = run(up, sync)[..., cols, :]
    abl(128); tail = run(up, sync)[..., cols, :] - bp.view(1, 32, 1, 1, 4)
    abl(0)
    for name, a, b in (("head part", head, head_ref), ("published part", tail, tail_ref)):
        d = (a - b).abs(); bad = (d > 1e-3).nonzero()
        print("unit %d %s: max err %.2e, bad %d" % (unit, name, d.max().item(), bad.shape[0]), end="")
        if bad.shape[0]: print("  rows", sorted(set(bad[:, 2].tolist())), "cols", sorted(set(bad[:, 3].tolist())), "groups", len(set(bad[:, 1].tolist())), end="")
        print()

import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cnmnet_amd import ops, _lib
lib = _lib.load(); dev = "cuda"
torch.manual_seed(0)
N, Cin, Cout, H, W = 1, 67, 128, 40, 72
x = ops.nchw_to_c4(torch.randn(N, Cin, H, W, device=dev)); w = torch.randn(Cout, 68, 3, 3, device=dev) * 0.05
up = ops.pack_winograd4(w); bp = torch.randn(Cout, device=dev)
lib.cnm_tune_wino36_staged(0); ref = ops.conv3x3_winograd4_c4(x, up, bp, Cout, False).clone()
lib.cnm_tune_wino36_staged(2)
SYNC = ops.wino36_sync_workspace(dev)
for rep in range(4):
    o = ops.conv3x3_winograd4_c4(x, up, bp, Cout, False, sync=SYNC).clone()
    torch.cuda.synchronize()
    d = (o - ref).abs()                       # [N, G, H, W, 4]
    bad = (d > 1e-3)
    # units: strips of 4 rows x 64 cols: sy = y//4, sx = x//64 ; unit = (sy*SW + sx)
    per = bad.any(dim=4).any(dim=1)[0]        # [H, W]
    units = {}
    for sy in range(10):
        for sx in range(2):
            blk = per[4*sy:4*sy+4, 64*sx:64*sx+64]
            units[(sy, sx)] = int(blk.sum())
    print("rep", rep, "bad elems", int(bad.sum()), "flags", float(SYNC[:1024].abs().max()), {k: v for k, v in units.items() if v})
# T = 20 units * 5 chunks = 100 phases, grid 25 -> ranges of 4: unit u = 5u..5u+4; cut units: all except those aligned

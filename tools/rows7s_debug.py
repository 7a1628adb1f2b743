"""Where does the split (sync workspace) path of the staged rows kernel differ from the unsplit one?  GPU box only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cnmnet_amd import ops, _lib
lib = _lib.load(); dev = "cuda"
sync = ops.wino36_sync_workspace(dev)
for (N, Cin, H, W) in [(1, 32, 4, 64), (1, 32, 4, 32 * 5), (1, 35, 8, 64), (2, 35, 37, 50)]:
    torch.manual_seed(1)
    x = ops.nchw_to_c4(torch.randn(N, Cin, H, W, device=dev)); wt = torch.randn(128, Cin, 7, 7, device=dev) * 0.02
    up = ops.pack_winograd(wt, stride=1, tile=4); bp = torch.randn(128, device=dev)
    RELU = os.environ.get("RELU", "1") == "1"; ref = ops.conv_rows_winograd_c4(x, up, bp, 128, 7, RELU, stride=1, tile=4)
    for rep in range(3):
        b = ops.conv_rows_winograd_c4(x, up, bp, 128, 7, RELU, stride=1, tile=4, sync=sync)
        d = (ref - b).abs()                                   # [N, G, H, W, 4]
        bad = (d > 1e-3).nonzero()
        ngrp = (Cin + 3) // 4; nch = (7 * ngrp + 7) // 8; SH = (H + 3) // 4; SW = (W + 31) // 32; nun = N * SH * SW; T = nun * nch; G = max(1, min(T // 4, 256))
        print("N%d Cin%d %dx%d: units %d phases/unit %d grid %d begins %s | max %.2e bad %d" % (N, Cin, H, W, nun, nch, G, [T * r // G for r in range(min(G, 8) + 1)], d.max().item(), bad.shape[0]), flush=True)
        if bad.shape[0]:
            n_, g_, y_, x_, e_ = [bad[:, i] for i in range(5)]
            units = (n_ * SH * SW + (y_ // 4) * SW + x_ // 32)
            print("   bad units:", sorted(set(units.tolist()))[:20], " groups:", sorted(set(g_.tolist()))[:40], " rows(in unit):", sorted(set((y_ % 4).tolist())), " cols(in unit):", sorted(set((x_ % 32).tolist()))[:40])
# which float4s of the published slots stayed zero?
sync2 = ops.wino36_sync_workspace(dev)
N, Cin, H, W = 1, 32, 4, 64
x = ops.nchw_to_c4(torch.randn(N, Cin, H, W, device=dev)); wt = torch.randn(128, Cin, 7, 7, device=dev) * 0.02
up = ops.pack_winograd(wt, stride=1, tile=4); bp = torch.randn(128, device=dev)
b = ops.conv_rows_winograd_c4(x, up, bp, 128, 7, True, stride=1, tile=4, sync=sync2); torch.cuda.synchronize()
slots = sync2[1024:].view(-1, 8, 1024, 4)                  # [range][wave][float4][4]
for r in range(3):
    z = (slots[r, :, :512].abs().sum(-1) == 0)             # [wave][512]
    idx = z.nonzero()
    print("slot of range %d: %d zero float4 of %d; (piece, lanes) =" % (r, idx.shape[0], 8 * 512), sorted(set((int(i) // 64, int(i) % 64) for _, i in idx.tolist()))[:40])

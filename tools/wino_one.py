"""Run one Winograd-kernel case a few times (for rocprofv3 --pmc): K in {3,5,7} via env K."""
import os, sys, torch
sys.path.insert(0, ".")
from cnmnet_amd import ops
K = int(os.environ.get("K", "7")); ALG = int(os.environ.get("ALG", "2"))
N, Cin, Cout, H, W = {7: (16, 67, 128, 192, 256), 5: (16, 128, 256, 96, 128), 3: (16, 256, 128, 96, 128)}[K]
dev = "cuda"
w = torch.randn(Cout, Cin, K, K, device=dev) * 0.02
x = ops.nchw_to_c4(torch.randn(N, Cin, H, W, device=dev))
up = ops.pack_winograd4(w) if ALG == 4 else ops.pack_winograd(w); bp = torch.zeros(Cout, device=dev)
for _ in range(5):
    y = (ops.conv3x3_winograd4_c4(x, up, bp, Cout, True, ksize=K) if ALG == 4 else ops.conv3x3_winograd_c4(x, up, bp, Cout, True)) if K == 3 or ALG == 4 else ops.conv_rows_winograd_c4(x, up, bp, Cout, K, True)
torch.cuda.synchronize()

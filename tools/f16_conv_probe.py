"""fp16 convolution layers at the bench shapes on the LDS-DMA kernel: every tile variant (cnm_tune_glds_tile) against the
automatic choice; results of the variants are checked against each other.  GPU box only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cnmnet_amd import _lib, ops

CASES = [  # (N, Cin, Cout, H, W, ksize, stride)
    (16, 67, 128, 192, 256, 7, 1), (16, 128, 128, 192, 256, 7, 2), (16, 128, 256, 96, 128, 5, 1), (16, 256, 256, 96, 128, 5, 2),
    (16, 256, 512, 48, 64, 3, 1), (16, 512, 512, 48, 64, 3, 2), (16, 512, 512, 24, 32, 3, 1), (16, 1024, 512, 24, 32, 3, 1),
    (16, 513, 256, 48, 64, 3, 1), (16, 257, 128, 96, 128, 3, 1), (16, 128, 64, 192, 256, 3, 1), (16, 65, 64, 192, 256, 3, 1),
    (16, 512, 512, 24, 32, 3, 2), (16, 512, 512, 12, 16, 3, 1), (16, 1024, 512, 12, 16, 3, 1), (16, 512, 512, 12, 16, 3, 2), (16, 512, 512, 6, 8, 3, 1),
    (8, 256, 256, 96, 128, 3, 2), (8, 256, 512, 48, 64, 3, 1), (8, 512, 512, 48, 64, 3, 2),
    (8, 67, 128, 192, 256, 3, 1), (8, 128, 128, 192, 256, 3, 2), (8, 512, 256, 48, 64, 3, 1), (8, 256, 128, 96, 128, 3, 1), (8, 64, 64, 192, 256, 3, 1),
]
NAMES = {0: "auto", 1: "128x256", 2: "64x512", 3: "64x128", 4: "128x512", 5: "256x256"}


def bench(fn, iters=10):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


lib = _lib.load(); dev = "cuda"
tot_auto = tot_best = 0.0
for N, Cin, Cout, H, W, k, st in CASES:
    w = torch.randn(Cout, Cin, k, k, device=dev) * 0.02
    x = ops.nchw_to_c8(torch.randn(N, Cin, H, W, device=dev))
    wp, bp = ops.pack_conv_f16(w, None, torch.randn(Cout, device=dev))
    fn = lambda: ops.conv2d_c8(x, wp, bp, Cout, k, st, True)
    res, ref = {}, None
    for v in ([1, 3, 4] + ([5] if Cout % 256 == 0 else []) + [0] if Cout % 128 == 0 else [2, 3, 0]):
        lib.cnm_tune_glds_tile(v)
        y = fn().float(); torch.cuda.synchronize()
        ref = y if ref is None else ref
        assert os.environ.get("F16_PROBE_NOCHECK") == "1" or float((y - ref).abs().max()) == 0.0, (v, float((y - ref).abs().max()))
        res[v] = bench(fn)
    lib.cnm_tune_glds_tile(0)
    lib.cnm_tune_gldsx(0); yo = fn().float(); res["old"] = bench(fn); lib.cnm_tune_gldsx(1)      # the automatic choice WITHOUT the row-extended kernel [r5]
    rowx_diff = float((yo - fn().float()).abs().max())
    gf = 2.0 * Cout * Cin * k * k * (H // st) * (W // st) * N / 1e9
    best = min((t, v) for v, t in res.items() if v and v != "old")
    tot_auto += res[0]; tot_best += best[0]; tot_old = globals().get("tot_old", 0.0) + res["old"]; globals()["tot_old"] = tot_old
    print("N%2d %4d->%4d k%d s%d %3dx%-3d: auto %.3f ms (%4.0f TF) | " % (N, Cin, Cout, k, st, H, W, res[0], gf / res[0]) +
          "  ".join("%s %.3f" % (NAMES[v], t) for v, t in res.items() if v and v != "old") + " | best %s | tap-by-tap kernel (gldsx=0) %.3f ms, max |difference| %.1e" % (NAMES[best[1]], res["old"], rowx_diff))
print("total auto %.2f ms, best-of forced tiles %.2f ms, automatic choice without the row-extended kernel %.2f ms" % (tot_auto, tot_best, tot_old))

"""Per-layer timing of the conv stack and K1 at a BASELINE configuration (GPU box only).
python tools/layer_sweep.py [--pairs 16] [--H 192] [--W 256]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cnmnet_amd import _lib, ops

DEPTH_LEVEL = [0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 4, 4, 3, 3, 2, 2, 1, 1, 0, 0]
REFINE_LEVEL = [0, 0, 1, 1, 2, 2, 2, 2, 1, 1, 0, 0, 2, 2, 1, 1, 0, 0]


def time_call(fn, iters=5, warm=2):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def sweep(net, N, H, W, planes=64):
    dev = torch.device("cuda:0")
    rows = []
    layers = [L for L in _lib.net_layers(net) if not L["is_head"]]
    levels = DEPTH_LEVEL if net == 0 else REFINE_LEVEL
    for L, lv in zip(layers, levels):
        cin = L["Cin"] if not (net == 0 and L["conv_key"] == "conv1.0") else 3 + planes
        h, w = H >> lv, W >> lv
        x = torch.randn(N, (cin + 3) // 4, h, w, 4, device=dev)
        wt = torch.randn(L["Cout"], cin, L["ksize"], L["ksize"], device=dev) * 0.01
        wp, bp = ops.pack_conv(wt)
        ms = time_call(lambda: ops.conv2d_c4(x, wp, bp, L["Cout"], L["ksize"], L["stride"], True))
        ho, wo = h // L["stride"], w // L["stride"]
        gflop = 2.0 * L["Cout"] * cin * L["ksize"] ** 2 * ho * wo * N / 1e9
        rows.append((L["conv_key"], cin, L["Cout"], L["ksize"], L["stride"], h, w, gflop, ms, gflop / ms))
    return rows


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=16); ap.add_argument("--H", type=int, default=192); ap.add_argument("--W", type=int, default=256)
    a = ap.parse_args()
    tot_g = tot_ms = 0
    for net, N in ((0, a.pairs), (1, a.pairs // 2)):
        print("net", net, "N", N)
        for r in sweep(net, N, a.H, a.W):
            print("%-18s cin %4d cout %4d k%d s%d %4dx%-4d %8.2f GFLOP %8.3f ms %7.1f TFLOP/s" % r)
            tot_g += r[7]; tot_ms += r[8]
    print("conv total %.1f GFLOP %.2f ms -> %.1f TFLOP/s" % (tot_g, tot_ms, tot_g / tot_ms))
    dev = torch.device("cuda:0")
    B, S, D = a.pairs // 2, 2, 64
    ref = torch.randn(B, 3, a.H, a.W, device=dev); src = torch.randn(B, S, 3, a.H, a.W, device=dev)
    from cnmnet_amd import synthetic as syn
    _, cams = syn.frames(B, S, a.H, a.W)
    cams = torch.from_numpy(cams).to(dev)
    hmkt = ops.homography_terms(cams[:, 0], cams[:, 1:])
    ms = time_call(lambda: ops.plane_sweep_cat_c4(ref, src, hmkt, 3.0, D), iters=20)
    byts = B * S * (3 * a.H * a.W * 4 * 2 + (D + 4) * a.H * a.W * 4)
    print("planesweep c4: %.3f ms  %.1f GB/s algorithmic" % (ms, byts / ms / 1e6))
    ms = time_call(lambda: ops.plane_sweep_volume(ref, src[:, 0], cams[:, 0], cams[:, 1], 3.0, D), iters=20)
    print("planesweep nchw (B pairs): %.3f ms" % ms)
    up = torch.randn(16, 32, 96, 128, 4, device=dev)
    ms = time_call(lambda: ops.upsample2x_c4(up)); print("upsample 128ch ->192x256 x16: %.3f ms, %.1f GB/s" % (ms, up.numel() * 4 * 5 / ms / 1e6))
    # fp16 conv stack (BASELINE config 5 shapes)
    tot_g = tot_ms = 0
    for net, N in ((0, a.pairs), (1, a.pairs // 2)):
        layers = [L for L in _lib.net_layers(net) if not L["is_head"]]
        for L, lv in zip(layers, DEPTH_LEVEL if net == 0 else REFINE_LEVEL):
            cin = L["Cin"] if not (net == 0 and L["conv_key"] == "conv1.0") else 67
            h, w = a.H >> lv, a.W >> lv
            x = torch.randn(N, (cin + 7) // 8, h, w, 8, device=dev).half()
            wp, bp = ops.pack_conv_f16(torch.randn(L["Cout"], cin, L["ksize"], L["ksize"], device=dev) * 0.01)
            ms = time_call(lambda: ops.conv2d_c8(x, wp, bp, L["Cout"], L["ksize"], L["stride"], True))
            gflop = 2.0 * L["Cout"] * cin * L["ksize"] ** 2 * (h // L["stride"]) * (w // L["stride"]) * N / 1e9
            print("f16 net%d %-18s cin %4d cout %4d k%d s%d %4dx%-4d %8.3f ms %7.1f TFLOP/s" % (net, L["conv_key"], cin, L["Cout"], L["ksize"], L["stride"], h, w, ms, gflop / ms))
            tot_g += gflop; tot_ms += ms
    print("f16 conv total %.1f GFLOP %.2f ms -> %.1f TFLOP/s" % (tot_g, tot_ms, tot_g / tot_ms))

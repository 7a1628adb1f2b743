#!/usr/bin/env python3
"""Every convolution layer of an fp16 step (depthNet on 2 x 8 images, DepthRefineNet on 8) alone: kernel instance, time, TFLOP/s
(direct-convolution flops).  The same loop as bench.py::f16_roofline, printed per layer instead of summed per kernel."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

import bench  # noqa: E402
from cnmnet_amd import _lib, ops  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    lib = _lib.load()
    for kv in os.environ.get("CNM_F16_TUNE", "").split(","):            # e.g. CNM_F16_TUNE=glds_deep=0,gldsx=0 (A/B in one call: the pool's boxes differ)
        if kv:
            k, v = kv.split("=")
            getattr(lib, "cnm_tune_" + k)(int(v))
    only = os.environ.get("CNM_F16_ONLY", "")                           # substring filter on "<net> <layer>"
    frames = 8
    tot = 0.0
    for net, n_img, levels, tag in ((_lib.NET_DEPTH, frames * bench.SRC, bench.DEPTH_LEVEL, "depth"), (_lib.NET_REFINE, frames, bench.REFINE_LEVEL, "refine")):
        layers = [L for L in _lib.net_layers(net) if not L["is_head"]]
        for L, lv in zip(layers, levels):
            if only and only not in ("%s %s %dx%d" % (tag, L["conv_key"], bench.H >> lv, bench.W >> lv)):
                continue
            cin = 3 + bench.PLANES if (net == _lib.NET_DEPTH and L["conv_key"] == "conv1.0") else L["Cin"]
            h, w = bench.H >> lv, bench.W >> lv
            k, st = L["ksize"], L["stride"]
            ho, wo = h // st, w // st
            wt = torch.randn(L["Cout"], cin, k, k, device=dev) * 0.02
            flop = 2.0 * L["Cout"] * cin * k * k * ho * wo * n_img
            g8 = (cin + 7) // 8
            if L["conv_key"].startswith("upconv") and cin <= 256 and n_img * ho * wo >= bench.UPSAMPLED_MIN_PIXELS_F16:
                xl = ops.nchw_to_c8(torch.randn(n_img, cin, h // 2, w // 2, device=dev))
                wp, bp, wr = ops.pack_upsampled_f16(wt)
                ms = bench.event_ms(lambda: ops.conv3x3_upsampled_c8(xl, wp, bp, L["Cout"], True, wr), iters=20, warm=3)
                name = "glds<%s,ups>+ring" % bench.glds_tile(4 * L["Cout"], n_img * (h // 2) * (w // 2), (9 * 8 * g8 + 63) // 64, 1)
            else:
                x = ops.nchw_to_c8(torch.randn(n_img, cin, h, w, device=dev))
                wp, bp = ops.pack_conv_f16(wt)
                ms = bench.event_ms(lambda: ops.conv2d_c8(x, wp, bp, L["Cout"], k, st, True), iters=20, warm=3)
                xt = bench.gldsx_tile(L["Cout"], cin, k, st, n_img, h, w)
                name = ("gldsx<%s>" % xt) if xt else "glds<%s>" % bench.glds_tile(L["Cout"], n_img * ho * wo, (k * k * 8 * g8 + 63) // 64, st)
            tot += ms
            print("%-6s %-18s N%2d %4d->%4d k%d s%d %3dx%-3d  %-28s %7.3f ms %7.1f TF" % (tag, L["conv_key"], n_img, cin, L["Cout"], k, st, h, w, name.replace(" ", ""), ms, flop / ms / 1e9), flush=True)
    print("sum %.3f ms" % tot)


if __name__ == "__main__":
    main()

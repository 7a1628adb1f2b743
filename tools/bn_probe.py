#!/usr/bin/env python3
"""BatchNorm (train mode) + ReLU at every layer shape of a training step (B = 4: depthNet on 8 images in 2 statistics groups, DepthRefineNet
on 4): forward (statistics + apply) and backward (reduce + apply) with HIP events, against the bytes each must move at 8 TB/s
(forward: read x twice, write y; backward: read x, dy twice each, write dx)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

import bench  # noqa: E402
from cnmnet_amd import _lib, autograd as ag  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    B = 4
    tf = tb = rf = rb = 0.0
    for net, n_img, S, levels, tag in ((_lib.NET_DEPTH, B * bench.SRC, bench.SRC, bench.DEPTH_LEVEL, "depth"), (_lib.NET_REFINE, B, 1, bench.REFINE_LEVEL, "refine")):
        layers = [L for L in _lib.net_layers(net) if not L["is_head"]]
        for L, lv in zip(layers, levels):
            C = L["Cout"]
            up = 2 if L["conv_key"].startswith("upconv") else 1
            h, w = (bench.H >> lv) // L["stride"] * 1, (bench.W >> lv) // L["stride"] * 1
            G = (C + 3) // 4
            x = torch.randn(n_img, G, h, w, 4, device=dev).requires_grad_(True)
            g = torch.ones(C, device=dev, requires_grad=True); b = torch.zeros(C, device=dev, requires_grad=True)
            rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
            dy = torch.randn_like(x)
            ys = []
            def fwd():
                ys.clear(); ys.append(ag.BatchNormReLUC4.apply(x, g, b, rm, rv, 0.1, 1e-5, True, None, S))
            ms_f = bench.event_ms(fwd, iters=10, warm=2)
            y = ys[0]
            def bwd():
                torch.autograd.grad(y, (x, g, b), dy, retain_graph=True)
            ms_b = bench.event_ms(bwd, iters=10, warm=2)
            nbytes = x.numel() * 4
            roof_f, roof_b = 3 * nbytes / 8e9, 5 * nbytes / 8e9        # ms at 8 TB/s
            tf += ms_f; tb += ms_b; rf += roof_f; rb += roof_b
            print("%-6s %-18s N%d C%4d %3dx%-3d %7.1f MB  fwd %6.1f us (%4.2f of roof)  bwd %6.1f us (%4.2f)" % (tag, L["conv_key"], n_img, C, h, w, nbytes / 1e6, ms_f * 1e3, roof_f / ms_f, ms_b * 1e3, roof_b / ms_b), flush=True)
    print("sum: forward %.3f ms (roof %.3f), backward %.3f ms (roof %.3f)" % (tf, rf, tb, rb))


if __name__ == "__main__":
    main()

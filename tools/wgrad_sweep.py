"""Per-layer timing of the weight-gradient kernel at the training shapes (B = 4 samples: 8 depthNet pairs / 4 refine frames,
192x256).  GPU box only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cnmnet_amd import _lib

DEPTH_LEVEL = [0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 4, 4, 3, 3, 2, 2, 1, 1, 0, 0]
REFINE_LEVEL = [0, 0, 1, 1, 2, 2, 2, 2, 1, 1, 0, 0, 2, 2, 1, 1, 0, 0]
lib = _lib.load(); dev = "cuda"


def time_call(fn, iters=5, warm=2):
    for _ in range(warm): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


tot_g = tot_ms = 0
for net, N in ((0, 4), (1, 4)):          # one depthNet call per source: 4 pairs each
    layers = [L for L in _lib.net_layers(net) if not L["is_head"]]
    for L, lv in zip(layers, DEPTH_LEVEL if net == 0 else REFINE_LEVEL):
        cin = L["Cin"] if not (net == 0 and L["conv_key"] == "conv1.0") else 67
        cout, k, st = L["Cout"], L["ksize"], L["stride"]
        h, w = 192 >> lv, 256 >> lv
        ho, wo = h // st, w // st
        G = (cin + 3) // 4
        x = torch.randn(N, G, h, w, 4, device=dev); dy = torch.randn(N, cout // 4, ho, wo, 4, device=dev)
        dw = torch.empty(cout, cin, k, k, device=dev)
        ws = torch.empty(lib.cnm_conv2d_wgrad_workspace_floats(cout, cin, k, N, ho, wo), device=dev)
        fn = lambda: _lib.check(lib.cnm_conv2d_wgrad_c4_f32(x.data_ptr(), G, 0, cin, dy.data_ptr(), cout // 4, 0, cout, dw.data_ptr(), ws.data_ptr(), ws.numel(),
                                                            N, h, w, k, st, 0, torch.cuda.current_stream().cuda_stream))
        ms = time_call(fn)
        gf = 2.0 * cout * cin * k * k * ho * wo * N / 1e9
        mult = 2 if net == 0 else 1
        tot_g += gf * mult; tot_ms += ms * mult
        print("net%d %-18s cin %4d cout %4d k%d s%d %4dx%-4d ws %6.1f MB %8.3f ms %6.1f TF" % (net, L["conv_key"], cin, cout, k, st, h, w, ws.numel() * 4 / 1e6, ms, gf / ms))
print("wgrad per step (depthNet x2 + refine): %.1f GFLOP %.2f ms -> %.1f TF" % (tot_g, tot_ms, tot_g / tot_ms))

"""Weight gradient of the 3x3 stride-1 layers: direct kernel (cnm_conv2d_wgrad_c4_f32) against the Winograd-domain one
(cnm_conv3x3_wgrad_winograd_c4_f32) at the training shapes -- relative L2 error against torch (fp64) and time (GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cnmnet_amd import _lib, ops
lib = _lib.load(); dev = "cuda"


def ms(fn, iters=10, warm=3):
    for _ in range(warm): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


st = lambda: torch.cuda.current_stream().cuda_stream
torch.manual_seed(0)
tot_d = tot_w = 0.0
for name, cin, cout, N, H, W, mult in (("iconv1", 65, 64, 4, 192, 256, 2), ("upconv1", 128, 64, 4, 192, 256, 4), ("refine conv1.0", 67, 128, 4, 192, 256, 1),
                                       ("iconv2", 257, 128, 4, 96, 128, 2), ("upconv2", 256, 128, 4, 96, 128, 4), ("conv2.0r", 128, 256, 4, 96, 128, 1),
                                       ("iconv3", 513, 256, 4, 48, 64, 2), ("conv3.0", 256, 512, 4, 48, 64, 3), ("upconv3", 512, 256, 4, 48, 64, 4),
                                       ("iconv4", 1024, 512, 4, 24, 32, 2), ("conv4.0", 512, 512, 4, 24, 32, 4), ("iconv5", 1024, 512, 4, 12, 16, 2), ("conv5.0", 512, 512, 4, 12, 16, 4)):
    G = (cin + 3) // 4
    x = torch.randn(N, cin, H, W, device=dev); dy = torch.randn(N, cout, H, W, device=dev)
    xc, dyc = ops.nchw_to_c4(x), ops.nchw_to_c4(dy)
    dwd, dww = torch.empty(cout, cin, 3, 3, device=dev), torch.empty(cout, cin, 3, 3, device=dev)
    wsd = torch.empty(lib.cnm_conv2d_wgrad_workspace_floats(cout, cin, 3, N, H, W), device=dev)
    wsw = torch.empty(lib.cnm_conv3x3_wgrad_winograd_workspace_floats(cout, cin, N, H, W), device=dev)
    fd = lambda: _lib.check(lib.cnm_conv2d_wgrad_c4_f32(xc.data_ptr(), G, 0, cin, dyc.data_ptr(), cout // 4, 0, cout, dwd.data_ptr(), wsd.data_ptr(), wsd.numel(), N, H, W, 3, 1, 0, st()))
    fw = lambda: _lib.check(lib.cnm_conv3x3_wgrad_winograd_c4_f32(xc.data_ptr(), G, 0, cin, dyc.data_ptr(), cout // 4, 0, cout, dww.data_ptr(), wsw.data_ptr(), wsw.numel(), N, H, W, 0, st()))
    fd(); fw(); torch.cuda.synchronize()
    if N * H * W <= 4 * 96 * 128:
        ref = torch.nn.grad.conv2d_weight(x.double(), (cout, cin, 3, 3), dy.double(), padding=1)
        rel = lambda a: float((a.double() - ref).norm() / ref.norm())
        err = "rel L2: direct %.1e  winograd %.1e" % (rel(dwd), rel(dww))
    else:
        err = "rel L2 between them %.1e" % float((dwd - dww).norm() / dwd.norm())
    td, tw = ms(fd), ms(fw)
    tot_d += td * mult; tot_w += tw * mult
    print("%-15s %4d->%4d %3dx%-3d N%d: direct %.3f ms  winograd %.3f ms  %.2fx   ws %5.0f MB   %s" % (name, cin, cout, H, W, N, td, tw, td / tw, wsw.numel() * 4 / 1e6, err), flush=True)
print("weighted by launches per step: direct %.2f ms, winograd %.2f ms" % (tot_d, tot_w))
for name, k, cin, cout, N, H, W in (("conv2.3", 5, 256, 256, 4, 96, 128), ("conv1.3", 7, 128, 128, 4, 192, 256), ("conv2.3 small", 5, 64, 128, 2, 24, 40), ("conv1.3 small", 7, 35, 64, 1, 26, 38)):
    G = (cin + 3) // 4
    x = torch.randn(N, cin, H, W, device=dev); dy = torch.randn(N, cout, H // 2, W // 2, device=dev)
    xc, dyc = ops.nchw_to_c4(x), ops.nchw_to_c4(dy)
    dwd, dww = torch.empty(cout, cin, k, k, device=dev), torch.full((cout, cin, k, k), 7.0, device=dev)
    wsd = torch.empty(lib.cnm_conv2d_wgrad_workspace_floats(cout, cin, k, N, H // 2, W // 2), device=dev)
    wsw = torch.empty(lib.cnm_conv_s2_wgrad_winograd_workspace_floats(cout, cin, k, N, H, W), device=dev)
    fd = lambda: _lib.check(lib.cnm_conv2d_wgrad_c4_f32(xc.data_ptr(), G, 0, cin, dyc.data_ptr(), cout // 4, 0, cout, dwd.data_ptr(), wsd.data_ptr(), wsd.numel(), N, H, W, k, 2, 0, st()))
    fw = lambda: _lib.check(lib.cnm_conv_s2_wgrad_winograd_c4_f32(xc.data_ptr(), G, 0, cin, dyc.data_ptr(), cout // 4, 0, cout, dww.data_ptr(), wsw.data_ptr(), wsw.numel(), N, H, W, k, 0, st()))
    fd(); fw(); torch.cuda.synchronize()
    ref = torch.nn.grad.conv2d_weight(x.double(), (cout, cin, k, k), dy.double(), stride=2, padding=k // 2)
    rel = lambda a_: float((a_.double() - ref).norm() / ref.norm())
    print("%-15s %dx%d s2 %4d->%4d %3dx%-3d N%d: direct %.3f ms  winograd %.3f ms  %.2fx   rel L2: direct %.1e  winograd %.1e" % (name, k, k, cin, cout, H, W, N, ms(fd), ms(fw), ms(fd) / ms(fw), rel(dwd), rel(dww)), flush=True)

for name, cin, cout, N, H, W, signed in (("conv1.0", 67, 128, 4, 192, 256, False), ("conv1.0 zero-mean x", 67, 128, 2, 48, 64, True), ("conv1.0 small", 35, 64, 1, 13, 30, False)):
    G = (cin + 3) // 4
    x = torch.randn(N, cin, H, W, device=dev); dy = torch.randn(N, cout, H, W, device=dev)
    if not signed: x = x.abs()
    xc, dyc = ops.nchw_to_c4(x), ops.nchw_to_c4(dy)
    dwd, dww = torch.empty(cout, cin, 7, 7, device=dev), torch.full((cout, cin, 7, 7), 7.0, device=dev)
    wsd = torch.empty(lib.cnm_conv2d_wgrad_workspace_floats(cout, cin, 7, N, H, W), device=dev)
    wsw = torch.empty(lib.cnm_conv7x7_wgrad_winograd_workspace_floats(cout, cin, N, H, W), device=dev)
    fd = lambda: _lib.check(lib.cnm_conv2d_wgrad_c4_f32(xc.data_ptr(), G, 0, cin, dyc.data_ptr(), cout // 4, 0, cout, dwd.data_ptr(), wsd.data_ptr(), wsd.numel(), N, H, W, 7, 1, 0, st()))
    fw = lambda: _lib.check(lib.cnm_conv7x7_wgrad_winograd_c4_f32(xc.data_ptr(), G, 0, cin, dyc.data_ptr(), cout // 4, 0, cout, dww.data_ptr(), wsw.data_ptr(), wsw.numel(), N, H, W, 0, st()))
    fd(); fw(); torch.cuda.synchronize()
    if N * H * W <= 2 * 48 * 64:
        ref = torch.nn.grad.conv2d_weight(x.double(), (cout, cin, 7, 7), dy.double(), padding=3)
        rel = lambda a_: float((a_.double() - ref).norm() / ref.norm())
        err = "rel L2: direct %.1e  winograd rows %.1e" % (rel(dwd), rel(dww))
    else:
        err = "rel L2 between them %.1e" % float((dwd - dww).norm() / dwd.norm())
    print("%-20s 7x7 %4d->%4d %3dx%-3d N%d: direct %.3f ms  winograd rows %.3f ms  %.2fx  ws %4.0f MB  %s" % (name, cin, cout, H, W, N, ms(fd), ms(fw), ms(fd) / ms(fw), wsw.numel() * 4 / 1e6, err), flush=True)

for name, cin, cout, N, H, W in (("conv2.0", 128, 256, 4, 96, 128), ("conv2.0 small", 36, 64, 2, 20, 30)):
    G = (cin + 3) // 4
    x = torch.randn(N, cin, H, W, device=dev); dy = torch.randn(N, cout, H, W, device=dev)
    xc, dyc = ops.nchw_to_c4(x), ops.nchw_to_c4(dy)
    dwd, dww = torch.empty(cout, cin, 5, 5, device=dev), torch.full((cout, cin, 5, 5), 7.0, device=dev)
    wsd = torch.empty(lib.cnm_conv2d_wgrad_workspace_floats(cout, cin, 5, N, H, W), device=dev)
    wsw = torch.empty(lib.cnm_conv5x5_wgrad_winograd_workspace_floats(cout, cin, N, H, W), device=dev)
    fd = lambda: _lib.check(lib.cnm_conv2d_wgrad_c4_f32(xc.data_ptr(), G, 0, cin, dyc.data_ptr(), cout // 4, 0, cout, dwd.data_ptr(), wsd.data_ptr(), wsd.numel(), N, H, W, 5, 1, 0, st()))
    fw = lambda: _lib.check(lib.cnm_conv5x5_wgrad_winograd_c4_f32(xc.data_ptr(), G, 0, cin, dyc.data_ptr(), cout // 4, 0, cout, dww.data_ptr(), wsw.data_ptr(), wsw.numel(), N, H, W, 0, st()))
    fd(); fw(); torch.cuda.synchronize()
    ref = torch.nn.grad.conv2d_weight(x.double(), (cout, cin, 5, 5), dy.double(), padding=2)
    rel = lambda a_: float((a_.double() - ref).norm() / ref.norm())
    print("%-20s 5x5 %4d->%4d %3dx%-3d N%d: direct %.3f ms  winograd rows %.3f ms  %.2fx  ws %4.0f MB  rel L2: direct %.1e  winograd rows %.1e" % (
        name, cin, cout, H, W, N, ms(fd), ms(fw), ms(fd) / ms(fw), wsw.numel() * 4 / 1e6, rel(dwd), rel(dww)), flush=True)

"""Probe: fused upsample + 3x3 conv (composed phase filters on the low-resolution input) vs upsample kernel + F(4x4,3x3)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cnmnet_amd import ops
import bench
dev = torch.device("cuda:0")
torch.manual_seed(0)
for (N, Cin, Cout, h, w) in ((16, 128, 64, 96, 128), (16, 256, 128, 48, 64), (16, 512, 256, 24, 32), (16, 512, 512, 12, 16), (8, 128, 64, 96, 128), (8, 256, 128, 48, 64), (8, 512, 256, 24, 32), (2, 64, 64, 9, 21), (1, 20, 128, 3, 40)):
    x = torch.randn(N, Cin // 4, h, w, 4, device=dev)
    wt = torch.randn(Cout, Cin, 3, 3, device=dev) * (2.0 / (9 * Cin)) ** 0.5
    bn = (torch.rand(Cout, device=dev) + 0.5, torch.randn(Cout, device=dev) * 0.1, torch.randn(Cout, device=dev) * 0.1, torch.rand(Cout, device=dev) + 0.5)
    u4 = ops.pack_winograd4(wt, bn); bp = ops.pack_conv(wt, bn)[1]
    uu, bu, wr = ops.pack_winograd4_upsampled(wt, bn)
    def old():
        return ops.conv3x3_winograd4_c4(ops.upsample2x_c4(x), u4, bp, Cout, True)
    def new():
        return ops.conv3x3_upsampled_winograd4_c4(x, uu, bu, Cout, True, wr)
    a, b = old(), new()
    d = (a - b).abs()
    inner = d[:, :, 1:-1, 1:-1].max().item(); ring = d.max().item()
    t_old = bench.event_ms(old, iters=10, warm=3); t_new = bench.event_ms(new, iters=10, warm=3)
    t_up = bench.event_ms(lambda: ops.upsample2x_c4(x), iters=10, warm=3)
    from cnmnet_amd import _lib
    lib = _lib.load(); P = lambda t_: t_.data_ptr(); st = torch.cuda.current_stream().cuda_stream
    outb = torch.empty(N, Cout // 4, 2 * h, 2 * w, 4, device=dev)
    t_ring = bench.event_ms(lambda: lib.cnm_conv3x3_upsampled_ring_c4_f32(P(x), Cin // 4, 0, Cin // 4, P(outb), Cout // 4, 0, Cout, P(wr), P(bu), N, h, w, 1, st), iters=10, warm=3)
    t_main1 = bench.event_ms(lambda: lib.cnm_conv3x3_upsampled_winograd4_c4_f32(P(x), Cin // 4, 0, Cin // 4, P(outb), Cout // 4, 0, Cout, P(uu), P(bu), N, h, w, 1, 1, st), iters=10, warm=3)
    print("   ring alone %.1f us, main(with_ring=1) %.1f us" % (t_ring * 1e3, t_main1 * 1e3))
    t_main = bench.event_ms(lambda: ops.conv3x3_upsampled_winograd4_c4(x, uu, bu, Cout, True), iters=10, warm=3)
    print("N%2d %3d->%3d %3dx%-3d  interior max|d| %.2e (ring %.2e, |out| %.2f)  old %.1f us (upsample %.1f)  fused %.1f us (main %.1f)" % (N, Cin, Cout, 2 * h, 2 * w, inner, ring, a.abs().max().item(), t_old * 1e3, t_up * 1e3, t_new * 1e3, t_main * 1e3))

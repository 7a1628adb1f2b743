"""LDS-staged 36-point Winograd kernel (conv_winograd4s.hip) against the gather-fed one (conv_winograd4.hip), GPU box only:
   (1) bit equality on a set of shapes (plain, concatenated input, fused up_conv, ragged edges, tile-block variants);
   (2) per-layer time of the F(4x4,3x3) launches of a bench step, both kernels interleaved in one process.
   python tools/wino36s_probe.py [check|time|all]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cnmnet_amd import ops, _lib

dev = "cuda"
lib = _lib.load()


SYNC = ops.wino36_sync_workspace(dev)


def both(fn):
    """gather-fed, staged with unit-aligned ranges, staged with the sync workspace (twice: reproducibility)"""
    outs = []
    for on, sync in ((0, None), (2, None), (2, SYNC), (2, SYNC)):
        lib.cnm_tune_wino36_staged(on)
        outs.append(fn(sync).clone())
    lib.cnm_tune_wino36_staged(1)
    assert ops.sync_workspace_state(SYNC) == 0, "workgroups left counted"
    return outs


def check():
    torch.manual_seed(0)
    bad = 0
    cases = [  # N, Cin, Cin2, Cout, H, W
        (2, 64, 0, 128, 48, 64), (1, 67, 0, 128, 40, 72), (3, 32, 0, 256, 24, 32), (2, 128, 129, 128, 32, 64), (1, 20, 0, 128, 24, 28),
        (2, 256, 0, 512, 48, 64), (1, 65, 0, 128, 192, 256), (5, 16, 0, 128, 8, 32), (1, 512, 513 - 512, 256, 16, 64), (2, 48, 0, 384, 52, 100),
        (4, 64, 0, 128, 12, 16), (3, 96, 0, 256, 14, 18), (16, 512, 0, 512, 12, 16)]
    for N, Cin, Cin2, Cout, H, W in cases:
        x = ops.nchw_to_c4(torch.randn(N, Cin, H, W, device=dev))
        x2 = ops.nchw_to_c4(torch.randn(N, Cin2, H, W, device=dev)) if Cin2 else None
        ct = 4 * ((Cin + 3) // 4) + Cin2
        w = torch.randn(Cout, ct, 3, 3, device=dev) * 0.05
        up = ops.pack_winograd4(w); bp = torch.randn(Cout, device=dev)
        o0, o1, o2, o3 = both(lambda sy: ops.conv3x3_winograd4_c4(x, up, bp, Cout, True, x2=x2, sync=sy))
        # independent check of the gather-fed result against torch (fp64) so that "equal" means "equal and right"
        xin = ops.c4_to_nchw(x, 4 * ((Cin + 3) // 4))
        if x2 is not None: xin = torch.cat([xin, ops.c4_to_nchw(x2, Cin2)], 1)
        ref = torch.relu(torch.nn.functional.conv2d(xin.double(), w.double(), bp.double(), padding=1)).float()
        err = (ops.c4_to_nchw(o1, Cout) - ref).abs().max().item()
        eq = torch.equal(o0, o1); rep = torch.equal(o2, o3)
        err2 = (ops.c4_to_nchw(o2, Cout) - ref).abs().max().item()
        bad += (not eq) or (not rep) or err > 2e-3 or err2 > 2e-3
        print("plain  N%d %4d+%-3d->%4d %3dx%-3d  aligned==gather %s  |aligned-torch64|=%.2e   split: reproducible %s  |split-torch64|=%.2e  |split-gather|=%.2e" % (
            N, Cin, Cin2, Cout, H, W, eq, err, rep, err2, (o0 - o2).abs().max().item()), flush=True)
    for N, Cin, Cout, H, W in [(2, 128, 64, 48, 64), (1, 256, 128, 24, 32), (2, 64, 64, 20, 36), (1, 128, 64, 96, 128)]:
        x = ops.nchw_to_c4(torch.randn(N, Cin, H, W, device=dev))
        w = torch.randn(Cout, Cin, 3, 3, device=dev) * 0.05
        uu, bu, wr = ops.pack_winograd4_upsampled(w)
        for ring in (None, wr):
            o0, o1, o2, o3 = both(lambda sy: ops.conv3x3_upsampled_winograd4_c4(x, uu, bu, Cout, True, ring, sync=sy))
            eq = torch.equal(o0, o1); rep = torch.equal(o2, o3); d2 = (o0 - o2).abs().max().item(); bad += (not eq) or (not rep) or d2 > 1e-3
            print("upconv N%d %4d->%4d %3dx%-3d ring=%d  aligned==gather %s   split: reproducible %s  |split-gather|=%.2e" % (N, Cin, Cout, H, W, ring is not None, eq, rep, d2), flush=True)
    for N, Cin, Cout, H, W in [(2, 64, 128, 40, 48), (1, 35, 256, 9, 29), (3, 128, 256, 24, 32), (1, 16, 128, 96, 128)]:     # F(2x2,5x5)
        x = ops.nchw_to_c4(torch.randn(N, Cin, H, W, device=dev))
        w = torch.randn(Cout, Cin, 5, 5, device=dev) * 0.03
        up = ops.pack_winograd4(w); bp = torch.randn(Cout, device=dev)
        o0, o1, o2, o3 = both(lambda sy: ops.conv3x3_winograd4_c4(x, up, bp, Cout, True, ksize=5, sync=sy))
        ref = torch.relu(torch.nn.functional.conv2d(ops.c4_to_nchw(x, 4 * ((Cin + 3) // 4))[:, :Cin].double(), w.double(), bp.double(), padding=2)).float()
        err = (ops.c4_to_nchw(o1, Cout) - ref).abs().max().item(); err2 = (ops.c4_to_nchw(o2, Cout) - ref).abs().max().item()
        eq = torch.equal(o0, o1); rep = torch.equal(o2, o3)
        bad += (not eq) or (not rep) or err > 2e-3 or err2 > 2e-3
        print("5x5    N%d %4d->%4d %3dx%-3d  aligned==gather %s  |aligned-torch64|=%.2e   split: reproducible %s  |split-torch64|=%.2e" % (N, Cin, Cout, H, W, eq, err, rep, err2), flush=True)
    print("CHECK", "FAILED" if bad else "OK", flush=True)
    return bad


SMALL = [("depth conv5.0", 16, 512, 512, 12, 16), ("depth iconv5", 16, 1024, 512, 12, 16), ("depth upconv5 (on 12x16)", 16, 512, 512, 12, 16)]
LAYERS = [  # (name, N, Cin, Cout, H, W)
    ("depth conv3.0", 16, 256, 512, 48, 64), ("depth conv4.0", 16, 512, 512, 24, 32), ("depth upconv3", 16, 512, 256, 48, 64), ("depth iconv3", 16, 513, 256, 48, 64),
    ("depth iconv2", 16, 257, 128, 96, 128), ("depth iconv1", 16, 65, 64, 192, 256), ("depth iconv4", 16, 1024, 512, 24, 32), ("depth upconv4", 16, 512, 512, 24, 32),
    ("refine conv1.0", 8, 67, 128, 192, 256), ("refine conv2.0", 8, 128, 256, 96, 128), ("refine conv3.0", 8, 256, 512, 48, 64),
    ("refine upconv3 x2", 8, 512, 256, 48, 64), ("refine iconv3 x2", 8, 512, 256, 48, 64), ("refine upconv2 x2", 8, 256, 128, 96, 128),
    ("refine iconv2 x2", 8, 256, 128, 96, 128), ("refine iconv1 x2", 8, 64, 64, 192, 256)]
UPS = [("depth upconv2 fused", 16, 256, 128, 48, 64), ("depth upconv1 fused", 16, 128, 64, 96, 128), ("refine upconv1 fused x2", 8, 128, 64, 96, 128)]


def ev(fn, iters=20, warm=3):
    for _ in range(warm): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def time_layers():
    tot = [0.0, 0.0, 0.0]
    for name, N, Cin, Cout, H, W in LAYERS:
        x = ops.nchw_to_c4(torch.randn(N, Cin, H, W, device=dev)); up = ops.pack_winograd4(torch.randn(Cout, Cin, 3, 3, device=dev) * 0.02); bp = torch.zeros(Cout, device=dev)
        ms = []
        for rnd in range(2):
            for on, sy in ((0, None), (2, None), (2, SYNC)):
                lib.cnm_tune_wino36_staged(on)
                m = ev(lambda: ops.conv3x3_winograd4_c4(x, up, bp, Cout, True, sync=sy))
                if rnd: ms.append(m)
        gf = 2.0 * Cout * Cin * 9 * H * W * N / 1e9
        mult = 2 if name.endswith("x2") else 1
        for i in range(3): tot[i] += ms[i] * mult
        print("%-24s %4d->%4d %3dx%-3d N%2d: gather %.3f ms (%.2f) | staged aligned %.3f ms (%.2f) x%.2f | staged split %.3f ms %5.1f TF (%.2f) x%.2f" % (
            name, Cin, Cout, H, W, N, ms[0], gf / ms[0] / 4 / 157.3, ms[1], gf / ms[1] / 4 / 157.3, ms[0] / ms[1], ms[2], gf / ms[2] / 4, gf / ms[2] / 4 / 157.3, ms[0] / ms[2]), flush=True)
    for name, N, Cin, Cout, H, W in SMALL:                                # F(2x2,3x3) kernel against the staged F(4x4,3x3) kernel with 4 x 4 tile blocks
        x = ops.nchw_to_c4(torch.randn(N, Cin, H, W, device=dev)); wt = torch.randn(Cout, Cin, 3, 3, device=dev) * 0.02
        u2, u4, bp = ops.pack_winograd(wt), ops.pack_winograd4(wt), torch.zeros(Cout, device=dev)
        lib.cnm_tune_wino36_staged(1)
        m2 = ev(lambda: ops.conv3x3_winograd_c4(x, u2, bp, Cout, True)); m4 = ev(lambda: ops.conv3x3_winograd4_c4(x, u4, bp, Cout, True, sync=SYNC))
        print("%-28s %4d->%4d %3dx%-3d N%2d: F(2x2) %.3f ms | staged F(4x4), 4x4 blocks %.3f ms  x%.2f" % (name, Cin, Cout, H, W, N, m2, m4, m2 / m4), flush=True)
    for name, N, Cin, Cout, H, W in UPS:
        x = ops.nchw_to_c4(torch.randn(N, Cin, H, W, device=dev)); uu, bu, wr = ops.pack_winograd4_upsampled(torch.randn(Cout, Cin, 3, 3, device=dev) * 0.02)
        ms = []
        for rnd in range(2):
            for on, sy in ((0, None), (2, None), (2, SYNC)):
                lib.cnm_tune_wino36_staged(on)
                m = ev(lambda: ops.conv3x3_upsampled_winograd4_c4(x, uu, bu, Cout, True, wr, sync=sy))
                if rnd: ms.append(m)
        gf = 2.0 * 4 * Cout * Cin * 9 * H * W * N / 1e9
        mult = 2 if name.endswith("x2") else 1
        for i in range(3): tot[i] += ms[i] * mult
        print("%-24s %4d->%4d %3dx%-3d N%2d: gather %.3f ms | staged aligned %.3f ms x%.2f | staged split %.3f ms %5.1f TF x%.2f   (with ring pass)" % (
            name, Cin, Cout, H, W, N, ms[0], ms[1], ms[0] / ms[1], ms[2], gf / ms[2] / 4, ms[0] / ms[2]), flush=True)
    for name, N, Cin, Cout, H, W in [("depth conv2.0 5x5", 16, 128, 256, 96, 128)]:
        x = ops.nchw_to_c4(torch.randn(N, Cin, H, W, device=dev)); up = ops.pack_winograd4(torch.randn(Cout, Cin, 5, 5, device=dev) * 0.02); bp = torch.zeros(Cout, device=dev)
        ms = []
        for rnd in range(2):
            for on, sy in ((0, None), (2, None), (2, SYNC)):
                lib.cnm_tune_wino36_staged(on)
                m = ev(lambda: ops.conv3x3_winograd4_c4(x, up, bp, Cout, True, ksize=5, sync=sy))
                if rnd: ms.append(m)
        gf = 2.0 * Cout * Cin * 25 * H * W * N / 1e9
        for i in range(3): tot[i] += ms[i]
        print("%-24s %4d->%4d %3dx%-3d N%2d: gather %.3f ms (%.2f) | staged aligned %.3f ms x%.2f | staged split %.3f ms %5.1f TF (%.2f) x%.2f" % (
            name, Cin, Cout, H, W, N, ms[0], gf * 0.36 / ms[0] / 157.3, ms[1], ms[0] / ms[1], ms[2], gf * 0.36 / ms[2], gf * 0.36 / ms[2] / 157.3, ms[0] / ms[2]), flush=True)
    lib.cnm_tune_wino36_staged(1)
    print("sum over a step's launches: gather %.3f ms, staged aligned %.3f ms, staged split %.3f ms" % tuple(tot), flush=True)


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    rc = 0
    if what in ("check", "all"): rc = check()
    if what in ("time", "all"): time_layers()
    sys.exit(1 if rc else 0)

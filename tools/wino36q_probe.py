"""Four-wave F(4x4,3x3) kernel (conv_winograd4q.hip) against the shipped kernels (eight-wave staged / gather-fed), GPU box only:
per-layer time of every 3x3 stride-1 layer of a bench step (N = 16 pairs depthNet, N = 8 refine), kernels interleaved in one process.
   python tools/wino36q_probe.py [reps]
Prints, per layer: shipped us, quad us, executed TF and fraction of the 157.3 TF fp32 matrix roof of both, and the winner."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cnmnet_amd import ops, _lib

dev = "cuda"
lib = _lib.load()
SYNC = ops.wino36_sync_workspace(dev)
REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 20

# name, N, Cin, Cin2, Cout, H, W, count per step
LAYERS = [
    ("d.conv3.0", 16, 256, 0, 512, 48, 64, 1), ("d.conv4.0", 16, 512, 0, 512, 24, 32, 1),
    ("d.upconv4(conv)", 16, 512, 0, 512, 24, 32, 1), ("d.iconv4", 16, 512, 512, 512, 24, 32, 1),
    ("d.upconv3(conv)", 16, 512, 0, 256, 48, 64, 1), ("d.iconv3", 16, 256, 257, 256, 48, 64, 1),
    ("d.iconv2", 16, 128, 129, 128, 96, 128, 1), ("d.iconv1", 16, 64, 1, 64, 192, 256, 1),
    ("r.conv1.0", 8, 67, 0, 128, 192, 256, 1), ("r.conv2.0", 8, 128, 0, 256, 96, 128, 1), ("r.conv3.0", 8, 256, 0, 512, 48, 64, 1),
    ("r.upconv3(conv)", 8, 512, 0, 256, 48, 64, 2), ("r.iconv3", 8, 256, 256, 256, 48, 64, 2),
    ("r.iconv2", 8, 128, 128, 128, 96, 128, 2), ("r.iconv1", 8, 32, 32, 64, 192, 256, 2),
]


def timed(fn, reps):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    torch.manual_seed(0)
    tot = {"shipped": 0.0, "quad": 0.0, "best": 0.0}
    for name, N, Cin, Cin2, Cout, H, W, cnt in LAYERS:
        x = ops.nchw_to_c4(torch.randn(N, Cin, H, W, device=dev))
        x2 = ops.nchw_to_c4(torch.randn(N, Cin2, H, W, device=dev)) if Cin2 else None
        ct = 4 * ((Cin + 3) // 4) + Cin2
        w = torch.randn(Cout, ct, 3, 3, device=dev) * 0.05
        up = ops.pack_winograd4(w); bp = torch.randn(Cout, device=dev)
        uq = ops.repack_winograd4_quad(up, Cout, ct)
        run_s = lambda: ops.conv3x3_winograd4_c4(x, up, bp, Cout, True, x2=x2, sync=SYNC)
        run_q = lambda: ops.conv3x3_winograd4q_c4(x, uq, bp, Cout, True, x2=x2, sync=SYNC)
        ok = bool(lib.cnm_conv3x3_winograd4q_ok(Cout, H, W))
        d = (run_s() - run_q()).abs().max().item() if ok else float("nan")
        ts, tq = [], []
        for _ in range(3):
            ts.append(timed(run_s, REPS))
            if ok:
                tq.append(timed(run_q, REPS))
        t_s, t_q = min(ts), (min(tq) if ok else float("inf"))
        tiles = N * ((H + 3) // 4) * ((W + 3) // 4)
        fl_s = tiles * 36 * Cout * (16 * ((4 * ((Cin + 3) // 4) + 4 * ((Cin2 + 3) // 4) + 15) // 16)) * 2      # executed: channels padded to the kernel's chunk
        fl_q = tiles * 36 * Cout * (8 * (((Cin + 3) // 4 + (Cin2 + 3) // 4 + 1) // 2)) * 2
        print("%-16s N%-2d %4d+%-3d->%3d %3dx%-3d  shipped %7.1f us %5.1f TF %.3f | quad %7.1f us %5.1f TF %.3f | quad/shipped %.3f  max|diff| %.1e" % (
            name, N, Cin, Cin2, Cout, H, W, t_s, fl_s / t_s / 1e6, fl_s / t_s / 1e6 / 157.3, t_q, fl_q / t_q / 1e6, fl_q / t_q / 1e6 / 157.3, t_q / t_s, d), flush=True)
        tot["shipped"] += cnt * t_s; tot["quad"] += cnt * (t_q if ok else t_s); tot["best"] += cnt * min(t_s, t_q)
    print("per step: shipped %.1f us, quad everywhere eligible %.1f us, best of both %.1f us" % (tot["shipped"], tot["quad"], tot["best"]))


if __name__ == "__main__":
    main()

"""How much of a kernel's time is the package power limit: the same launches on random and on all-zero operands (zero operands toggle
fewer wires: less power, higher sustained clock), with rocm-smi sampled meanwhile.  python tools/power_data_probe.py   (GPU box)"""
import os, re, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cnmnet_amd import ops
dev = "cuda"


def sample_while(fn, seconds=2.0):
    got = []
    def smp():
        time.sleep(0.5)
        for _ in range(3):
            o = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=15).stdout
            m, p = re.search(r"sclk clock level: \d+: \((\d+)Mhz\)", o), re.search(r"Power \(W\): ([\d.]+)", o)
            if m: got.append((int(m.group(1)), float(p.group(1)) if p else 0.0))
    th = threading.Thread(target=smp); th.start()
    n = 0; torch.cuda.synchronize(); t0 = time.perf_counter()
    while th.is_alive():
        for _ in range(50): fn()
        torch.cuda.synchronize(); n += 50
    dt = (time.perf_counter() - t0) / n
    return dt * 1e6, sum(g[0] for g in got) / max(len(got), 1), sum(g[1] for g in got) / max(len(got), 1)


def case(name, make):
    for kind in ("random", "zeros"):
        fn = make(kind == "zeros")
        us, mhz, w = sample_while(fn)
        print("%-44s %-6s %8.1f us per launch   sclk %4.0f MHz   %4.0f W" % (name, kind, us, mhz, w), flush=True)


def f32(zero):
    x = torch.zeros(16, 64, 48, 64, 4, device=dev) if zero else torch.randn(16, 64, 48, 64, 4, device=dev)
    w = torch.zeros(512, 256, 3, 3, device=dev) if zero else torch.randn(512, 256, 3, 3, device=dev) * 0.02
    up, bp, sync = ops.pack_winograd4(w), torch.zeros(512, device=dev), ops.wino36_sync_workspace(dev)
    return lambda: ops.conv3x3_winograd4_c4(x, up, bp, 512, True, sync=sync)


def f16(zero):
    x = ops.nchw_to_c8(torch.zeros(16, 128, 96, 128, device=dev) if zero else torch.randn(16, 128, 96, 128, device=dev))
    w = torch.zeros(256, 128, 5, 5, device=dev) if zero else torch.randn(256, 128, 5, 5, device=dev) * 0.02
    wp, bp = ops.pack_conv_f16(w, None, torch.zeros(256, device=dev))
    return lambda: ops.conv2d_c8(x, wp, bp, 256, 5, 1, True)


case("fp32 staged F(4x4,3x3), 256->512 @48x64 x16", f32)
case("fp16 implicit GEMM 256x256, 128->256 5x5 @96x128 x16", f16)

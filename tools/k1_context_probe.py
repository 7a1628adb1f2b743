#!/usr/bin/env python3
"""The plane-sweep launch of the headline shape in different neighbourhoods on ONE box: what precedes the launch decides part of its duration
(the pool's boxes differ by 20 % on this kernel inside the step and by 1 % on the convolutions).  Every launch is timed by the library's own
fence-free event pair around the kernel (cnm_debug_sweep_timing_arm / _read)."""
import ctypes
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

from cnmnet_amd import _lib, ops, synthetic as syn  # noqa: E402

dev = torch.device("cuda:0")
B, S, Hh, Ww, D = 8, 2, 192, 256, 64
img, cams = syn.frames(B, S, Hh, Ww, seed=77)
img, cams = torch.from_numpy(img).to(dev), torch.from_numpy(cams).to(dev)
ref, src = img[:, 0].contiguous(), img[:, 1:].contiguous()
hmkt = ops.homography_terms(cams[:, 0], cams[:, 1:])
lib = _lib.load()
ws = torch.zeros(lib.cnm_planesweep_workspace_floats(B, S, Hh, Ww), device=dev)
out = ops.plane_sweep_cat_c4(ref, src, hmkt, 3.0, D, ws=ws)
lo, hi = ops.idepth_range(3.0)
args = (ref.data_ptr(), src.data_ptr(), hmkt.data_ptr(), out.data_ptr(), ws.data_ptr(), ws.numel(), B, S, Hh, Ww, D, lo, hi, torch.cuda.current_stream().cuda_stream)
wt = torch.randn(128, 3 + D, 7, 7, device=dev) * 0.02
up, (_, bp) = ops.pack_winograd(wt, stride=1), ops.pack_conv(wt)
x512 = torch.randn(16, 128, 24, 32, 4, device=dev)
w512, b512 = ops.pack_winograd4(torch.randn(512, 512, 3, 3, device=dev) * 0.02), torch.zeros(512, device=dev)
sync = ops.wino36_sync_workspace(dev)
small = torch.zeros(64, device=dev)


def heavy():
    ops.conv_rows_winograd_c4(out, up, bp, 128, 7, True, stride=1)          # conv1.0, 2 ms of fp32 MFMA


big_a = torch.randn(100 * 1024 * 1024, device=dev); big_b = torch.empty_like(big_a)     # 400 MB each


def dirty():
    big_b.copy_(big_a)                                                       # 400 MB written: the memory-side cache full of another buffer's dirty lines, little power


def matrix_small():
    for _ in range(10):                                                      # ~1.4 ms of fp32 MFMA on 12.6 MB tensors: power, no footprint
        ops.conv3x3_winograd4_c4(x512, w512, b512, 512, relu=True, sync=sync)


def tiny():
    ops.homography_terms(cams[:, 0], cams[:, 1:])                            # one small workgroup, ~6 us


def run(before, n=40, gap_s=0.0):
    buf = (ctypes.c_float * n)()
    lib.cnm_debug_sweep_timing_arm(n)
    for _ in range(n):
        before()
        if gap_s:
            torch.cuda.synchronize(); time.sleep(gap_s)
        _lib.check(lib.cnm_planesweep_cat_c4_f32(*args))
    torch.cuda.synchronize()
    got = lib.cnm_debug_sweep_timing_read(buf, n)
    v = sorted(buf[i] * 1e3 for i in range(got))[2:]
    return "median %.1f  p10 %.1f  p90 %.1f us" % (v[len(v) // 2], v[len(v) // 10], v[9 * len(v) // 10])


for rep in range(2):
    print("after conv1.0 (2 ms of matrix work)          :", run(heavy))
    print("after conv1.0 + one tiny kernel               :", run(lambda: (heavy(), tiny())))
    print("after conv1.0 + three tiny kernels            :", run(lambda: (heavy(), tiny(), tiny(), tiny())))
    print("after a 400 MB copy (dirty lines, no matrix)  :", run(dirty))
    print("after 1.4 ms of matrix work on 12 MB tensors  :", run(matrix_small))
    print("back to back (the previous sweep's write-back):", run(lambda: None))
    print("after an idle device (sync + 2 ms sleep)      :", run(lambda: None, n=20, gap_s=0.002))
    print("after an idle device, behind one tiny kernel  :", run(tiny, n=20, gap_s=0.002))

#!/usr/bin/env python3
"""Sustained HBM WRITE bandwidth against buffer size (torch fill_): below the 256 MB memory-side cache a rewritten buffer never reaches HBM; above it
every byte does.  The plane sweep writes 210 MB per launch: alone it lives in the first regime, inside a step (the cache full of other kernels' dirty
lines) in the second."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

import bench  # noqa: E402

dev = torch.device("cuda:0")
for mb in (64, 128, 210, 256, 384, 512, 1024, 2048, 4096):
    x = torch.empty(mb * 1024 * 1024 // 4, device=dev)
    ms = bench.event_ms(lambda: x.fill_(1.0), iters=10, warm=3)
    y = torch.empty_like(x)
    ms_c = bench.event_ms(lambda: y.copy_(x), iters=10, warm=3) if mb <= 2048 else None
    print("%5d MB  fill %8.1f us = %6.0f GB/s written%s" % (mb, ms * 1e3, mb * 1.048576 / ms, "" if ms_c is None else "   copy %8.1f us = %6.0f GB/s read + as much written" % (ms_c * 1e3, mb * 1.048576 / ms_c)), flush=True)
    del x, y

#!/usr/bin/env python3
"""Per-CU reading of a tools/k1_bench.hip -DSWEEP_TRACE dump (gpurun_out/k1_trace.txt): unit durations by queue zone, how long a
CU has two / one / no workgroup inside a unit, when CUs finish.  usage: tools/k1_trace_report.py [file] [grid]"""
import sys
import numpy as np
from collections import defaultdict
fn = sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/k1_trace.txt'
grid = int(sys.argv[2]) if len(sys.argv) > 2 else 512
d = np.loadtxt(fn, dtype=np.int64)
wg, unit, hw, xcc, t0r, tf, t1r = d.T
base = t0r.min(); M = (1 << 44) - 1
t0 = (t0r - base) / 100.0
t1 = ((t1r - (base & M)) & M) / 100.0
tf = tf / 100.0
cuid = xcc * 1000 + ((hw >> 13) & 7) * 100 + ((hw >> 12) & 1) * 50 + ((hw >> 8) & 0xf)
dur = t1 - t0
print(f'{len(d)} units on {len(set(cuid))} CUs; first entry -> last end {t1.max():.1f} us')
for lo in range(0, int(unit.max()) + 1, grid // 2):
    m = (unit >= lo) & (unit < lo + grid // 2)
    if m.any():
        print(f'  units {lo:4d}+: n={m.sum():3d} start {t0[m].min():5.1f} {np.median(t0[m]):5.1f} {t0[m].max():5.1f} | duration {dur[m].min():5.1f} {np.median(dur[m]):5.1f} {dur[m].max():5.1f} | to the footprint barrier {np.median(tf[m]):4.1f} (max {tf[m].max():4.1f}) | end {np.median(t1[m]):5.1f} {t1[m].max():5.1f}')
percu = defaultdict(list)
for i in range(len(d)):
    percu[cuid[i]].append((t0[i], t1[i]))
two, one, ends = [], [], []
for v in percu.values():
    ev = sorted([(a, 1) for a, _ in v] + [(b, -1) for _, b in v])
    n = 0; last = 0.0; t2 = 0.0; t1_ = 0.0
    for t, dl in ev:
        if n >= 2: t2 += t - last
        elif n == 1: t1_ += t - last
        n += dl; last = t
    two.append(t2); one.append(t1_); ends.append(max(b for _, b in v))
print(f'  per CU (median): two workgroups in a unit {np.median(two):.1f} us, one {np.median(one):.1f} us; last end min {min(ends):.1f} median {np.median(ends):.1f} max {max(ends):.1f}')

#!/bin/bash
# Durations of the plane-sweep launches INSIDE the benchmark step (one per step, between the convolution kernels),
# from a rocprofv3 kernel trace of bench.py without the roofline bursts (GPU box).
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
rm -rf /tmp/k1is; timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/k1is -- python3 bench.py --steps 50 --warmup 10 --no-roofline --no-cpu-baseline --no-secondary --no-live-traffic > /dev/null 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob("/tmp/k1is/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(((int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3) for r in csv.DictReader(open(f)) if "planesweep_kernel<1," in r["Kernel_Name"]))
import os
if os.environ.get("K1_SERIES"): print("in launch order:", " ".join("%.0f" % v for _, v in rows))
if os.environ.get("K1_NEIGHBOURS"):
    allr = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:46]) for r in csv.DictReader(open(f))))
    shown = 0
    for i, (s0, e0, nm) in enumerate(allr):
        if "planesweep_kernel<1," in nm and 2 <= i < len(allr) - 2 and shown < 8 and i > len(allr) // 2:
            shown += 1
            print("  ...", " | ".join("%s %.1f us (gap %.1f)" % (allr[j][2][:28], (allr[j][1] - allr[j][0]) / 1e3, (allr[j][0] - allr[j - 1][1]) / 1e3) for j in range(i - 2, i + 2)))
d = sorted(v for _, v in rows)
n = len(d)
print("plane sweep inside the step: %d launches, min %.1f  p10 %.1f  median %.1f  mean %.1f  p90 %.1f  max %.1f us -> 224.9 MB / median = %.0f GB/s = %.3f of 8 TB/s" % (
    n, d[0], d[n // 10], d[n // 2], sum(d) / n, d[9 * n // 10], d[-1], 224.919552 / d[n // 2] * 1e3, 224.919552 / d[n // 2] / 8.0))
PY

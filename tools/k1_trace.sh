#!/bin/bash
# Per-launch durations of the plane-sweep kernel (rocprofv3 kernel trace) in the standalone harness (GPU box).
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize tools/k1_bench.hip -o /tmp/k1_st 2>/dev/null
rm -rf /tmp/k1st; timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/k1st -- /tmp/k1_st 8 192 256 64 stats 1 0 > /dev/null 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob("/tmp/k1st/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "planesweep_kernel<1>" in r["Kernel_Name"]]
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
gap = [(int(rows[i + 1]["Start_Timestamp"]) - int(rows[i]["End_Timestamp"])) / 1e3 for i in range(len(rows) - 1)]
print("durations (us):", " ".join("%.0f" % x for x in d))
print("gaps (us):", " ".join("%.1f" % x for x in gap))
PY

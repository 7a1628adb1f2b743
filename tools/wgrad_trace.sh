#!/bin/bash
# Every conv_wgrad_kernel dispatch of one training step (GPU box): grid, duration -- from a rocprofv3 kernel trace of tools/train_bench.py.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
rm -rf /tmp/wgt; timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/wgt -- python3 tools/train_bench.py > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("/tmp/wgt/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted((r for r in csv.DictReader(open(f))), key=lambda r: int(r["Start_Timestamp"]))
w = [r for r in rows if "conv_wgrad_kernel" in r["Kernel_Name"]]
per_step = len(w) // 18
last = w[-per_step:]
agg = collections.OrderedDict()
for r in last:
    key = (r["Kernel_Name"].split("(")[0][-24:], int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"]))
    agg.setdefault(key, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = 0.0
for (k, x, y, z), v in agg.items():
    tot += sum(v)
    print("%-24s grid %2d x %2d x %4d = %5d workgroups  x%d  %7.1f us each" % (k, x, y, z, x * y * z, len(v), sum(v) / len(v)))
print("sum %.2f ms over %d launches of the last step" % (tot / 1e3, per_step))
PY

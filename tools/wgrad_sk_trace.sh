#!/bin/bash
# [r6] per-kernel totals of the weight-gradient path in one training step, split form against stream-K (rocprofv3 kernel trace of tools/train_bench.py).
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
for v in 0 1; do
  rm -rf /tmp/wgt$v; CNM_WGRAD_STREAMK=$v timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/wgt$v -- python3 tools/train_bench.py > /dev/null 2>&1
  python3 - $v <<'PY'
import csv, glob, collections, sys
v = sys.argv[1]
f = glob.glob("/tmp/wgt%s/**/*kernel_trace.csv" % v, recursive=True)[0]
rows = sorted((r for r in csv.DictReader(open(f))), key=lambda r: int(r["Start_Timestamp"]))
names = ("conv_wgrad", "wgrad_reduce", "wino_wgrad_finish", "wino_wgrad_rows_finish", "wino_wgrad_xform", "wino_wgrad_rows_xform", "fillBuffer", "memset")
agg = collections.OrderedDict()
nstep = 18
for r in rows:
    n = r["Kernel_Name"]
    for k in names:
        if k in n:
            key = n.split("(")[0][:60]
            a = agg.setdefault(key, [0, 0.0]); a[0] += 1; a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
            break
tot = 0.0
print("== wgrad_streamk=%s" % v)
for k, (c, us) in agg.items():
    tot += us
    print("%-62s x%5.1f/step %8.1f us/step" % (k, c / nstep, us / nstep))
print("sum %.2f ms per step" % (tot / nstep / 1e3))
if v == "1":
    w = [r for r in rows if "conv_wgrad_sk" in r["Kernel_Name"]]
    per = len(w) // nstep
    for r in w[-per:]:
        print("   %-28s grid %5d  %8.1f us" % (r["Kernel_Name"].split("(")[0][-28:], int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
PY
done

#!/bin/bash
# [r6] What the `train` step (normal / warped-depth losses, Depth2normal k = 9) adds to `train_wo_normal`: per-kernel totals per step of both, and the kernels that differ.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
for m in wo normals; do
  rm -rf /tmp/tn_$m
  if [ $m = wo ]; then a=""; else a="normals"; fi
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tn_$m -- python3 tools/train_bench.py 4 $a > /dev/null 2>&1
done
python3 - <<'PY'
import csv, glob
def load(m):
    f = glob.glob("/tmp/tn_%s/**/*kernel_stats.csv" % m, recursive=True)[0]
    return {r["Name"]: (int(r["Calls"]) / 18.0, int(r["TotalDurationNs"]) / 18e3) for r in csv.DictReader(open(f))}
a, b = load("wo"), load("normals")
print("total us/step: wo_normal %.0f, train %.0f" % (sum(v[1] for v in a.values()), sum(v[1] for v in b.values())))
rows = []
for k in set(a) | set(b):
    ca, ua = a.get(k, (0, 0)); cb, ub = b.get(k, (0, 0))
    if abs(ub - ua) > 5: rows.append((ub - ua, k, ca, ua, cb, ub))
for d, k, ca, ua, cb, ub in sorted(rows, reverse=True)[:40]:
    print("%+8.0f us  %-90s  %5.1f x %7.0f us -> %5.1f x %7.0f us" % (d, k[:90], ca, ua, cb, ub))
PY

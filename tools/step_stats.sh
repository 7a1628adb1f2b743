#!/bin/bash
# Per-kernel totals of one bench configuration under rocprofv3 (GPU box):  tools/step_stats.sh <tag> [bench args...]
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
tag=$1; shift
O=gpurun_out/stats_$tag; rm -rf $O; mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-live-traffic --no-roofline "$@" > $O/bench.json 2> $O/bench.err
f=$(find $O -name '*kernel_stats.csv' | head -1)
cp "$f" gpurun_out/kernel_stats_$tag.csv
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = 25.0
tot = 0
for r in rows[:16]:
    per = float(r['TotalDurationNs']) / steps / 1e6
    tot += per
    print("%-70s calls/step %5.1f avg %8.1f us  per step %6.3f ms" % (r['Name'][:70], int(r['Calls']) / steps, float(r['AverageNs']) / 1e3, per))
print("sum of all kernels per step: %.3f ms" % (sum(float(r['TotalDurationNs']) for r in rows) / steps / 1e6))
PY
tail -c 300 $O/bench.json | head -c 200; echo
rm -rf $O

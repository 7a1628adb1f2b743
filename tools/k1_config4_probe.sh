#!/bin/bash
# [r6] where the plane sweep's time goes at config 4 (640 x 480, 96 planes, 16 pairs) next to the headline size: prebuilt tools/bin/k1_*.bin
# (tools/k1_bench.hip with -DSWEEP_STATS / -DSWEEP_SPAN / nothing), no host check (ncheck 0).
cd "$(dirname "$0")/.."
for sz in "8 192 256 64" "8 480 640 96" "8 480 640 64" "8 384 512 96" "8 192 256 96"; do
  echo "=== B H W D = $sz"
  for v in plain stats span; do timeout 300 tools/bin/k1_$v.bin $sz "$v" 1 0 2>&1 | grep -v "launches ok\|launch ok\|check:"; done
done

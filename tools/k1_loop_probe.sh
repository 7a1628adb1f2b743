#!/bin/bash
# Build and run tools/k1_loop_probe.hip in its modes, at 8 and 4 waves per SIMD (GPU box).
cd "$(dirname "$0")/.."
for minw in 8 4; do for ahead in 1 0; do for m in 0 1 2 3 4; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -DSWEEP_MINW=$minw -DSWEEP_AHEAD=$ahead -DPROBE_MODE=$m tools/k1_loop_probe.hip -o /tmp/k1lp 2>/dev/null || { echo build failed; continue; }
  echo -n "[MINW=$minw AHEAD=$ahead] "; /tmp/k1lp
  [ $m -gt 0 ] && [ $ahead = 0 ] && true
done; done; done

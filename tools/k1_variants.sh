#!/bin/bash
# Build and time plane-sweep kernel variants on the GPU box: tools/k1_variants.sh "<-D flags>" ...
cd "$(dirname "$0")/.."
i=0
for flags in "$@"; do
  i=$((i+1))
  hipcc --offload-arch=gfx950 -O3 -std=c++17 $flags tools/k1_bench.hip -o /tmp/k1_$i 2>/dev/null || { echo "build failed: $flags"; continue; }
  /tmp/k1_$i 8 192 256 64 "[$flags]"
done

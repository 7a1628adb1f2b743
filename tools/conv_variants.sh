#!/bin/bash
# Build and time conv kernel variants on the GPU box: tools/conv_variants.sh "<-D flags>" ...
cd "$(dirname "$0")/.."
i=0
for flags in "$@"; do
  i=$((i+1))
  hipcc --offload-arch=gfx950 -O3 -std=c++17 $flags tools/conv_bench.hip -o /tmp/cb_$i 2>/tmp/cb_err_$i || { echo "build failed: $flags"; head -5 /tmp/cb_err_$i; continue; }
  /tmp/cb_$i -1 "[$flags]"
done

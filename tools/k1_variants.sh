#!/bin/bash
# Build and time plane-sweep kernel variants on the GPU box:
#   tools/k1_variants.sh "<-D flags>[@B H W D geom ncheck]" ...
set -u
cd "$(dirname "$0")/.."
i=0
for spec in "$@"; do
  i=$((i+1))
  flags="${spec%%@*}"; args="8 192 256 64"; rest="1 2"
  if [[ "$spec" == *@* ]]; then a=(${spec#*@}); args="${a[0]} ${a[1]} ${a[2]} ${a[3]}"; rest="${a[4]:-1} ${a[5]:-2}"; fi
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize $flags tools/k1_bench.hip -o /tmp/k1_$i 2>/tmp/k1_build_$i.log || { echo "build failed: $flags"; tail -5 /tmp/k1_build_$i.log; continue; }
  timeout 300 /tmp/k1_$i $args "[$spec]" $rest || echo "run failed ($?): $spec"
done

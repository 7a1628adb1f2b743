#!/bin/bash
# The plane sweep INSIDE the benchmark step, for several builds of planesweep.hip (GPU box): recompile that one object, relink, trace.
#   tools/k1_in_step_variants.sh "<-D flags>" ...
cd "$(dirname "$0")/.."
L=cnmnet_amd/lib
for flags in "$@" ""; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize $flags -c cnmnet_amd/csrc/planesweep.hip -o $L/planesweep.o 2>/dev/null || { echo "build failed: $flags"; continue; }
  hipcc --offload-arch=gfx950 -shared -fPIC -pthread $(ls $L/*.o | grep -v _cblk0) -o $L/libcnm_engine.so
  echo -n "[$flags] "; tools/k1_in_step.sh
done

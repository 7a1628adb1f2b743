#!/bin/bash
# Build libcnm_engine variants of the Winograd kernel with extra -D flags and time them (GPU box).
# usage: tools/wino_variants.sh "name:-DFLAG ..." ...
set -e
cd "$(dirname "$0")/.."
for v in "$@"; do
  name="${v%%:*}"; flags="${v#*:}"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $flags -c cnmnet_amd/csrc/conv_winograd.hip -o /tmp/wv.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $flags -c cnmnet_amd/csrc/conv_winograd_rows.hip -o /tmp/wr.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $flags -c cnmnet_amd/csrc/conv_winograd4.hip -o /tmp/w4.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC cnmnet_amd/lib/planesweep.o cnmnet_amd/lib/conv_mfma.o /tmp/wv.o /tmp/wr.o /tmp/w4.o cnmnet_amd/lib/pointwise.o cnmnet_amd/lib/geometry.o cnmnet_amd/lib/nets.o cnmnet_amd/lib/train_ops.o cnmnet_amd/lib/half_ops.o -o cnmnet_amd/lib/libcnm_engine.so
  echo "== $name ($flags)"
  python tools/wino_check.py 2>&1 | grep -E "^N16 " | head -${NLINES:-3} | awk '{print $1,$2,$3, $(NF-5), $(NF-4), $(NF-3), $(NF-2), $NF}'
done

#!/bin/bash
# [r6] Is the weight-gradient GEMM waiting for its operands?  A twin library built with -DWGRAD_SAMEDATA (every tile reads tile (0, 0) of problem 0:
# the operand stream of a launch fits the L2; wrong results) against the product library, per-kernel totals of a training step.
# `build` (CPU container), then `run` (GPU box).
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
L=cnmnet_amd/lib
if [ "$1" = build ]; then
  objs="$L/planesweep.o $L/conv_mfma.o $L/conv_winograd.o $L/conv_winograd4.o $L/conv_winograd4s.o $L/conv_winograd4q.o $L/conv_winograd_rows.o $L/conv_rows_staged.o $L/pointwise.o $L/geometry.o $L/nets.o $L/half_ops.o $L/host_twins.o"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -DWGRAD_SAMEDATA ${WGRAD_EXTRA:-} -c cnmnet_amd/csrc/train_ops.hip -o $L/train_ops_same.o || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -pthread $objs $L/train_ops_same.o -o $L/libcnm_engine_same.so && ls -la $L/libcnm_engine_same.so
  exit
fi
for lib in libcnm_engine.so libcnm_engine_same.so; do
  echo "== $lib"
  rm -rf /tmp/wgs; CNM_ENGINE_LIB=$PWD/$L/$lib timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/wgs -- python3 tools/train_bench.py > /dev/null 2>&1
  f=$(find /tmp/wgs -name '*kernel_stats.csv' | head -1)
  grep -E "conv_wgrad" "$f" | awk -F'","' '{printf "%-60s calls %5s total %9.3f ms avg %8.1f us\n", substr($1,2,58), $2, $3/1e6, $4/1e3}'
done

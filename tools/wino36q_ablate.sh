#!/bin/bash
# Ablation timings of the four-wave F(4x4,3x3) kernel.  `build` (CPU container): an -DWINO4Q_ABLATE twin of the library,
# cnmnet_amd/lib/libcnm_engine_qabl.so (git-ignored, travels with the tree); `run` (GPU box): the modes on three layer shapes.
# masks: 1 no input transform, 2 no raw loads / stores, 4 no weight loads, 8 no B-fragment reads, 16 no output transform
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
L=cnmnet_amd/lib
if [ "$1" = build ]; then
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -DWINO4Q_ABLATE $WQ_FLAGS -c cnmnet_amd/csrc/conv_winograd4q.hip -o $L/conv_winograd4q_abl.o || exit 1
  objs="$L/planesweep.o $L/conv_mfma.o $L/conv_winograd.o $L/conv_winograd4.o $L/conv_winograd4s.o $L/conv_winograd_rows.o $L/conv_rows_staged.o $L/pointwise.o $L/geometry.o $L/nets.o $L/train_ops.o $L/half_ops.o $L/host_twins.o"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -pthread $objs $L/conv_winograd4q_abl.o -o $L/libcnm_engine_qabl.so && echo built $L/libcnm_engine_qabl.so
  exit
fi
export CNM_ENGINE_LIB=$PWD/$L/libcnm_engine_qabl.so
for S in "256 512 48 64 16" "257 128 96 128 16" "65 64 192 256 16"; do
  for m in 0 1 2 4 8 16 3 7 15 31; do timeout 120 python3 tools/wino36q_one.py $S $m 20 2>&1 | grep quad; done
done

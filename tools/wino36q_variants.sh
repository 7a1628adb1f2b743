#!/bin/bash
# Schedule variants of the four-wave kernel's input transform (WQ_VS / WQ_VR / WQ_WS / WQ_WR, conv_winograd4q.hip).
# `build` (CPU container): one -DWINO4Q_ABLATE twin library per variant; `run` (GPU box): full kernel and no-transform on two layers.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
L=cnmnet_amd/lib
if [ "${WQ_SWEEP:-1}" = 3 ]; then VARIANTS=("60 1 64 76 -DWINO4Q_WD=18" "60 1 64 76 -DWINO4Q_WD=9" "60 1 64 76 -DWINO4Q_WD=6" "60 1 64 76 -DWINO4Q_WD=3")   # third sweep: weight fragments in flight
elif [ "${WQ_SWEEP:-1}" = 2 ]; then VARIANTS=("60 1 64 76" "80 1 84 56" "100 1 104 36" "30 1 34 100" "60 1 64 36" "90 1 94 46")   # second sweep: where the clump sits
else VARIANTS=("44 1 48 90" "24 114 0 1" "44 6 52 86" "44 1 48 36" "60 1 64 76" "40 2 48 90"); fi
if [ "$1" = build ]; then
  objs="$L/planesweep.o $L/conv_mfma.o $L/conv_winograd.o $L/conv_winograd4.o $L/conv_winograd4s.o $L/conv_winograd_rows.o $L/conv_rows_staged.o $L/pointwise.o $L/geometry.o $L/nets.o $L/train_ops.o $L/half_ops.o $L/host_twins.o"
  i=0
  for v in "${VARIANTS[@]}"; do
    set -- $v
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -DWINO4Q_ABLATE -DWQ_VS=$1 -DWQ_VR=$2 -DWQ_WS=$3 -DWQ_WR=$4 ${5:-} -c cnmnet_amd/csrc/conv_winograd4q.hip -o $L/conv_winograd4q_qv$i.o &
    i=$((i+1))
  done
  wait
  for ((j=0;j<i;j++)); do /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -pthread $objs $L/conv_winograd4q_qv$j.o -o $L/libcnm_engine_qv$j.so; done
  ls -la $L/libcnm_engine_qv*.so
  exit
fi
i=0
for v in "${VARIANTS[@]}"; do
  echo "== variant $i: VS VR WS WR [flags] = $v"
  export CNM_ENGINE_LIB=$PWD/$L/libcnm_engine_qv$i.so
  for S in "256 512 48 64 16" "65 64 192 256 16"; do
    for m in 0 1; do timeout 120 python3 tools/wino36q_one.py $S $m 20 2>&1 | grep quad; done
  done
  i=$((i+1))
done

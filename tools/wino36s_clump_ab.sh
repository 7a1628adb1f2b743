#!/bin/bash
# [r6] the eight-wave staged kernel with its transform in two clumps (-DWINO4S_CLUMP=1) against the shipped spread, alternating on one box.
# `build` (CPU container): twin libraries; `run` (GPU box): tools/wino36s_probe.py time + a short bench for each, twice.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
L=cnmnet_amd/lib
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -mllvm -greedy-regclass-priority-trumps-globalness=1 -mllvm -disable-machine-licm"
VARIANTS=("-DWINO4S_CLUMP=1" "-DWINO4S_CLUMP=1 -DWINO4S_CLUMP_A=5 -DWINO4S_CLUMP_B=11" "-DWINO4S_CLUMP=1 -DWINO4S_CLUMP_A=3 -DWINO4S_CLUMP_B=9")
if [ "$1" = build ]; then
  objs="$L/planesweep.o $L/conv_mfma.o $L/conv_winograd.o $L/conv_winograd4.o $L/conv_winograd4q.o $L/conv_winograd_rows.o $L/conv_rows_staged.o $L/pointwise.o $L/geometry.o $L/nets.o $L/train_ops.o $L/half_ops.o $L/host_twins.o"
  i=0
  for v in "${VARIANTS[@]}"; do
    /opt/rocm/bin/hipcc $F $v -Rpass-analysis=kernel-resource-usage -c cnmnet_amd/csrc/conv_winograd4s.hip -o $L/conv_winograd4s_cl$i.o 2> /tmp/cl$i.log &
    i=$((i+1))
  done
  wait
  for ((j=0;j<i;j++)); do grep -A9 "conv_winograd36s_f32_kernelILi16ELb0ELi0ELi4ELb0" /tmp/cl$j.log | grep -E "VGPRs:|Spill|Scratch" | tr '\n' ' '; echo; /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -pthread $objs $L/conv_winograd4s_cl$j.o -o $L/libcnm_engine_cl$j.so; done
  ls $L/libcnm_engine_cl*.so
  exit
fi
one() {
  echo "== $1"
  timeout 300 python3 tools/wino36s_probe.py time 2>&1 | tail -1
  timeout 300 python3 bench.py --steps 30 --warmup 5 --no-roofline --no-secondary --no-cpu-baseline --no-live-traffic 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   bench', round(d['value'],1), 'frames/s', round(d['ms_per_step'],3), 'ms')"
}
for rep in 1 2; do
  unset CNM_ENGINE_LIB; one "shipped"
  i=0
  for v in "${VARIANTS[@]}"; do export CNM_ENGINE_LIB=$PWD/$L/libcnm_engine_cl$i.so; one "clump: $v"; i=$((i+1)); done
done

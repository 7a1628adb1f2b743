#!/bin/bash
# [r6] Where the weight-gradient GEMM's time goes: twin libraries with -DWGRAD_ABL=<bits> (1 no global loads, 2 no LDS stores, 4 no barriers,
# 8 no fragment reads; wrong results), the kernels' totals over a training step.  `build` (CPU container), then `run` (GPU box).
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
L=cnmnet_amd/lib
MODES="${WGRAD_MODES:-0 1 3 7 15 8}"
if [ "$1" = build ]; then
  objs="$L/planesweep.o $L/conv_mfma.o $L/conv_winograd.o $L/conv_winograd4.o $L/conv_winograd4s.o $L/conv_winograd4q.o $L/conv_winograd_rows.o $L/conv_rows_staged.o $L/pointwise.o $L/geometry.o $L/nets.o $L/half_ops.o $L/host_twins.o"
  for m in $MODES; do /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -DWGRAD_ABL=$m ${WGRAD_EXTRA:-} -c cnmnet_amd/csrc/train_ops.hip -o $L/train_ops_abl$m.o & done; wait
  for m in $MODES; do /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -pthread $objs $L/train_ops_abl$m.o -o $L/libcnm_engine_abl$m.so; done
  ls -la $L/libcnm_engine_abl*.so; exit
fi
for m in $MODES; do
  echo "== WGRAD_ABL=$m"
  rm -rf /tmp/wga; CNM_ENGINE_LIB=$PWD/$L/libcnm_engine_abl$m.so timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/wga -- python3 tools/train_bench.py > /dev/null 2>&1
  python3 - <<'PY'
import csv, glob
f = glob.glob("/tmp/wga/**/*kernel_stats.csv", recursive=True)[0]
tot = 0.0
for r in csv.DictReader(open(f)):
    if "conv_wgrad" in r["Name"]:
        tot += int(r["TotalDurationNs"]) / 18e6
        print("   %-50s x%5.1f/step %8.3f ms/step  avg %7.1f us" % (r["Name"].split("(")[0][:50], int(r["Calls"]) / 18, int(r["TotalDurationNs"]) / 18e6, float(r["AverageNs"]) / 1e3))
print("   sum %.3f ms per step" % tot)
PY
done

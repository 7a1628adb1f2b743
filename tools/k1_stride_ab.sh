#!/bin/bash
# [r6] the plane sweep with frame strides in its arguments against a -DSWEEP_DENSE twin (the addressing of rounds 1-5), launch timed inside the bench step.
# `build` (CPU container), then `run` (GPU box; CNM_FRAME_VIEWS=0 so that both libraries see dense inputs).
cd "$(dirname "$0")/.."
L=cnmnet_amd/lib
if [ "$1" = build ]; then
  objs="$L/conv_mfma.o $L/conv_winograd.o $L/conv_winograd4.o $L/conv_winograd4s.o $L/conv_winograd4q.o $L/conv_winograd_rows.o $L/conv_rows_staged.o $L/pointwise.o $L/geometry.o $L/nets.o $L/train_ops.o $L/half_ops.o $L/host_twins.o"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -fno-slp-vectorize -DSWEEP_DENSE -c cnmnet_amd/csrc/planesweep.hip -o $L/planesweep_dense.o || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -pthread $objs $L/planesweep_dense.o -o $L/libcnm_engine_dense.so && ls -la $L/libcnm_engine_dense.so
  exit
fi
for rep in 1 2 3; do
  for lib in libcnm_engine_dense.so libcnm_engine.so; do
    echo -n "$lib: "
    CNM_FRAME_VIEWS=0 CNM_ENGINE_LIB=$PWD/$L/$lib python3 bench.py --no-cpu-baseline --no-secondary --no-live-traffic 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['roofline_planesweep']; print('%.1f frames/s, sweep in the step %.2f us (frac %.3f), policy %s' % (d['value'], p['avg_launch_ms']*1e3, p['frac'], p.get('store_policy',{}).get('in_force')))"
  done
done

#!/bin/bash
# [r6] the up_conv ring pass with eight waves per workgroup (shipped) against four (-DRING_NW=4): the pass alone on the step's shapes, fp32 and fp16
# steps.  `build` here (twin library), `run` on the GPU box, alternating.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
L=cnmnet_amd/lib
if [ "$1" = build ]; then
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -DRING_NW=4 -c cnmnet_amd/csrc/conv_winograd4.hip -o $L/conv_winograd4_nw4.o || exit 1
  objs="$L/planesweep.o $L/conv_mfma.o $L/conv_winograd.o $L/conv_winograd4s.o $L/conv_winograd4q.o $L/conv_winograd_rows.o $L/conv_rows_staged.o $L/pointwise.o $L/geometry.o $L/nets.o $L/train_ops.o $L/half_ops.o $L/host_twins.o"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -pthread $objs $L/conv_winograd4_nw4.o -o $L/libcnm_engine_nw4.so && echo built
  exit
fi
one() {
  echo "== $1"
  python3 - <<'EOF' 2>/dev/null
import torch, bench
from cnmnet_amd import ops, _lib
lib = _lib.load(); P = lambda t: t.data_ptr(); st = torch.cuda.current_stream().cuda_stream
for (N, Cin, Cout, h, w) in ((16, 128, 64, 96, 128), (16, 256, 128, 48, 64), (8, 128, 64, 96, 128)):
    x = torch.randn(N, Cin // 4, h, w, 4, device="cuda"); wt = torch.randn(Cout, Cin, 3, 3, device="cuda") * 0.05
    uu, bu, wr = ops.pack_winograd4_upsampled(wt)
    out = torch.empty(N, Cout // 4, 2 * h, 2 * w, 4, device="cuda")
    t = bench.event_ms(lambda: lib.cnm_conv3x3_upsampled_ring_c4_f32(P(x), Cin // 4, 0, Cin // 4, P(out), Cout // 4, 0, Cout, P(wr), P(bu), N, h, w, 1, st), iters=20, warm=3)
    print("   ring pass N%d %d->%d low-res %dx%d: %.1f us" % (N, Cin, Cout, h, w, t * 1e3))
EOF
  for prec in f32 f16; do
    timeout 300 python3 bench.py --precision $prec --steps 30 --warmup 5 --no-roofline --no-secondary --no-cpu-baseline --no-live-traffic 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   bench $prec', round(d['value'],1), 'frames/s', round(d['ms_per_step'],3), 'ms')"
  done
}
for rep in 1 2; do
  unset CNM_ENGINE_LIB; one "eight waves (shipped)"
  export CNM_ENGINE_LIB=$PWD/$L/libcnm_engine_nw4.so; one "four waves (round 5)"
done

#!/bin/bash
# [r6] where the fused up_conv instance of the eight-wave kernel (conv_winograd36s<16, true>: 0.57 of the matrix roof) spends its time:
# ablation modes of an -DWINO4S_ABLATE twin library on the step's fused up_conv shapes.  `build` here, `run` on the GPU box.
# masks: 1 no input transform, 2 no DMA, 4 no weight loads, 16 no output transform / stores, 7 = 1 + 2 + 4
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
L=cnmnet_amd/lib
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -mllvm -greedy-regclass-priority-trumps-globalness=1 -mllvm -disable-machine-licm"
if [ "$1" = build ]; then
  /opt/rocm/bin/hipcc $F -DWINO4S_ABLATE -c cnmnet_amd/csrc/conv_winograd4s.hip -o $L/conv_winograd4s_uabl.o || exit 1
  objs="$L/planesweep.o $L/conv_mfma.o $L/conv_winograd.o $L/conv_winograd4.o $L/conv_winograd4q.o $L/conv_winograd_rows.o $L/conv_rows_staged.o $L/pointwise.o $L/geometry.o $L/nets.o $L/train_ops.o $L/half_ops.o $L/host_twins.o"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -pthread $objs $L/conv_winograd4s_uabl.o -o $L/libcnm_engine_uabl.so && echo built
  exit
fi
export CNM_ENGINE_LIB=$PWD/$L/libcnm_engine_uabl.so
python3 - <<'EOF'
import ctypes, torch
from cnmnet_amd import ops, _lib
lib = _lib.load()
f = ctypes.CDLL(_lib.LIB_PATH).cnm_tune_wino36s_ablate; f.argtypes = [ctypes.c_int]
sync = ops.wino36_sync_workspace("cuda")
for name, N, Cin, Cout, H, W in (("d.upconv2", 16, 256, 128, 48, 64), ("d.upconv1", 16, 128, 64, 96, 128), ("r.upconv2", 8, 256, 128, 48, 64), ("r.upconv1", 8, 128, 64, 96, 128)):
    x = ops.nchw_to_c4(torch.randn(N, Cin, H, W, device="cuda"))
    uu, bu, wr = ops.pack_winograd4_upsampled(torch.randn(Cout, Cin, 3, 3, device="cuda") * 0.05)
    out = torch.empty(N, Cout // 4, 2 * H, 2 * W, 4, device="cuda")
    for m in (0, 16, 1, 2, 4, 7):
        f(m)
        fn = lambda: ops.conv3x3_upsampled_winograd4_c4(x, uu, bu, Cout, True, None, out=out, sync=sync)
        for _ in range(3): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        gf = 2.0 * 4 * Cout * Cin * 36 * ((H + 3) // 4) * ((W + 3) // 4) * N / 1e9
        print("%-10s N%-2d %3d->%3d low-res %2dx%-3d ablate %2d: %.4f ms  %.1f TF executed (%.3f of 157.3)" % (name, N, Cin, Cout, H, W, m, ms, gf / ms, gf / ms / 157.3), flush=True)
    f(0)
    try:
        ops.engine_status(clear=True)
    except Exception:
        pass
EOF

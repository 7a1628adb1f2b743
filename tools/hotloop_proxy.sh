#!/bin/bash
# CPU-side codegen proxy for the staged 36-point kernel: instruction mix of the dominant instance's per-phase hot path
# (144 MFMAs; spill reloads = v_readlane / scratch_load inside it are what small source changes move).  usage: tools/hotloop_proxy.sh [-D flags]
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -S --cuda-device-only -DWINO4S_ONE_INSTANCE "$@" -o /tmp/hotloop_proxy.s cnmnet_amd/csrc/conv_winograd4s.hip 2>/dev/null && python3 tools/hotloop_mix.py /tmp/hotloop_proxy.s

#!/bin/bash
# Build-time variants of the staged 36-point kernel: the sum line of tools/wino36s_probe.py for each (GPU box).
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
for v in "" "-DWINO4S_DMA_SPREAD=4" "-DWINO4S_EARLY_BARRIER=0" "-DWINO4S_DMA_SPREAD=4 -DWINO4S_EARLY_BARRIER=0" "-DWINO4S_DMA_SPREAD=3" "-DWINO4S_DMA_SPREAD=1 -DWINO4S_EARLY_BARRIER=0"; do
  rm -f cnmnet_amd/lib/conv_winograd4s.o
  CNM_EXTRA_HIPCC_FLAGS="$v" python3 -m cnmnet_amd.build > /tmp/build.log 2>&1 || { tail -3 /tmp/build.log; continue; }
  echo "== variant [$v]"
  for i in 1 2; do timeout 300 python3 tools/wino36s_probe.py 2>&1 | tail -1; done
done
rm -f cnmnet_amd/lib/conv_winograd4s.o

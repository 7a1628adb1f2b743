#!/bin/bash
# Ablation timings of the LDS-staged Winograd kernel (GPU box; rebuilds conv_winograd4s.o with -DWINO4S_ABLATE).
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
for hs in 6 4; do
  rm -f cnmnet_amd/lib/conv_winograd4s.o
  CNM_EXTRA_HIPCC_FLAGS="-DWINO4S_ABLATE -DWINO4S_HI_STEP=$hs" python3 -m cnmnet_amd.build > /dev/null 2>&1
  echo "== hi-wave DMA step $hs"
  for L in "256 512 48 64 16" "67 128 192 256 8" "257 128 96 128 16"; do
    for m in 0 2 16 18 4 1; do timeout 120 python3 tools/wino36s_one.py $L 1 $m 30 2>&1 | grep staged; done
  done
done
rm -f cnmnet_amd/lib/conv_winograd4s.o

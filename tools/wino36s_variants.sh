#!/bin/bash
# Build-time variants of the staged 36-point kernel: the sum line of tools/wino36s_probe.py for each (GPU box).
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
for v in "-DWINO4S_LO_STEP=14 -DWINO4S_LO_SHARE=7" "-DWINO4S_LO_STEP=10 -DWINO4S_LO_SHARE=7" "-DWINO4S_LO_STEP=12 -DWINO4S_LO_SHARE=7" "-DWINO4S_LO_STEP=16 -DWINO4S_LO_SHARE=7" ""; do
  rm -f cnmnet_amd/lib/conv_winograd4s.o
  CNM_EXTRA_HIPCC_FLAGS="$v" python3 -m cnmnet_amd.build > /tmp/build.log 2>&1 || { tail -3 /tmp/build.log; continue; }
  echo "== variant [$v]"
  for i in 1; do timeout 300 python3 tools/wino36s_probe.py 2>&1 | tail -1; done
done
rm -f cnmnet_amd/lib/conv_winograd4s.o

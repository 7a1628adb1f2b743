#!/bin/bash
# Weight-prefetch depth sweep of the LDS-staged Winograd kernel (GPU box: rebuilds conv_winograd4s.o per depth).
cd "$(dirname "$0")/.."
L="${1:-256 512 48 64 16}"
for wd in 2 4 6; do
  rm -f cnmnet_amd/lib/conv_winograd4s.o
  CNM_EXTRA_HIPCC_FLAGS="-DWINO4S_WD=$wd" python3 -m cnmnet_amd.build > /dev/null 2>&1
  for r in 1 2; do echo -n "WD=$wd  "; timeout 120 python3 tools/wino36s_one.py $L 1 0 30 2>&1 | grep staged; done
done
rm -f cnmnet_amd/lib/conv_winograd4s.o

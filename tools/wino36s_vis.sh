#!/bin/bash
cd "$(dirname "$0")/.."
for fl in "" "-DWINO4S_FENCE" "-DWINO4S_SC1LOAD=16" "-DWINO4S_SC1LOAD=17" "-DWINO4S_FENCE -DWINO4S_SC1LOAD=17"; do
  rm -f cnmnet_amd/lib/conv_winograd4s.o
  CNM_EXTRA_HIPCC_FLAGS="$fl" python3 -m cnmnet_amd.build > /dev/null 2>&1
  echo "== flags: $fl"; timeout 120 python3 tools/wino36s_debug.py 2>&1 | grep "^rep" | cut -c1-150
done
rm -f cnmnet_amd/lib/conv_winograd4s.o

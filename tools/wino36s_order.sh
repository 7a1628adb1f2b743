#!/bin/bash
# Unit order A/B of the staged kernel (GPU box: rebuilds conv_winograd4s.o per setting).
cd "$(dirname "$0")/.."
for o in 0 1 0 1; do
  rm -f cnmnet_amd/lib/conv_winograd4s.o
  CNM_EXTRA_HIPCC_FLAGS="-DWINO4S_CBLK_SLOW=$o" python3 -m cnmnet_amd.build > /dev/null 2>&1
  echo "== channel block slowest = $o"
  timeout 300 python3 tools/wino36s_probe.py time 2>&1 | grep -v "^/opt" | awk '{print $1, $2, $3, $4, $5, $6, "split", $(NF-6), $(NF-5), $(NF-2), $(NF-1), $NF}' | tail -20
done
rm -f cnmnet_amd/lib/conv_winograd4s.o

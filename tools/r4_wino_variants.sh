#!/bin/bash
# Round 4: build-time variants of the staged 36-point kernel, the sum line of tools/wino36s_probe.py time for each (GPU box).
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
for v in "" "-DWINO4S_PRIO=1" "-DWINO4S_PRIO=2" "-DWINO4S_CBLK_SLOW=0" ""; do
  rm -f cnmnet_amd/lib/conv_winograd4s.o
  CNM_EXTRA_HIPCC_FLAGS="$v" python3 -m cnmnet_amd.build > /tmp/build.log 2>&1 || { tail -3 /tmp/build.log; continue; }
  echo "== variant [$v]"
  timeout 300 python3 tools/wino36s_probe.py time 2>&1 | tail -3
done
rm -f cnmnet_amd/lib/conv_winograd4s.o
python3 -m cnmnet_amd.build > /tmp/build.log 2>&1

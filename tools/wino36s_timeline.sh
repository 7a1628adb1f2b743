#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
rm -f cnmnet_amd/lib/conv_winograd4s.o
CNM_EXTRA_HIPCC_FLAGS="-DWINO4S_TIMELINE $WINO4S_FLAGS" python3 -m cnmnet_amd.build > /tmp/build.log 2>&1 || tail -5 /tmp/build.log
timeout 200 python3 tools/wino36s_timeline.py 256 512 48 64 16 2>&1 | tail -24
timeout 200 python3 tools/wino36s_timeline.py 128 128 192 256 8 2>&1 | tail -24
rm -f cnmnet_amd/lib/conv_winograd4s.o

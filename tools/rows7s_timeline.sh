#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
rm -f cnmnet_amd/lib/conv_rows_staged.o
CNM_EXTRA_HIPCC_FLAGS="-DROWS7S_TIMELINE $ROWS7S_FLAGS" python3 -m cnmnet_amd.build > /tmp/build.log 2>&1 || tail -5 /tmp/build.log
timeout 200 python3 tools/rows7s_timeline.py 2>&1 | tail -30
rm -f cnmnet_amd/lib/conv_rows_staged.o

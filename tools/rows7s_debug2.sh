#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
rm -f cnmnet_amd/lib/conv_rows_staged.o
CNM_EXTRA_HIPCC_FLAGS="-DROWS7S_ABLATE" python3 -m cnmnet_amd.build > /dev/null 2>&1
timeout 200 python3 tools/rows7s_debug2.py 2>&1 | tail
timeout 200 python3 tools/rows7s_debug.py 2>&1 | grep -v "bad units\|slot of" | tail -12
rm -f cnmnet_amd/lib/conv_rows_staged.o

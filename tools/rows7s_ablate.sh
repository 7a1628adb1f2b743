#!/bin/bash
# Ablation timings of the staged 7x7 rows kernel (GPU box; rebuilds conv_rows_staged.o with -DROWS7S_ABLATE).
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
rm -f cnmnet_amd/lib/conv_rows_staged.o
CNM_EXTRA_HIPCC_FLAGS="-DROWS7S_ABLATE" python3 -m cnmnet_amd.build > /dev/null 2>&1
for m in 0 1 2 3 4 8 16 32 7 15 31 47; do timeout 120 python3 tools/rows7s_one.py $m 0 2>&1 | grep staged; done
rm -f cnmnet_amd/lib/conv_rows_staged.o

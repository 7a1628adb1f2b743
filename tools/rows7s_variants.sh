#!/bin/bash
# Build-time variants of the staged 7x7 rows kernel, one conv1.0-shaped launch each (GPU box).
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
for v in "" "-DROWS7S_NOBARRIER"; do
  rm -f cnmnet_amd/lib/conv_rows_staged.o
  CNM_EXTRA_HIPCC_FLAGS="$v" python3 -m cnmnet_amd.build > /tmp/build.log 2>&1 || { tail -3 /tmp/build.log; continue; }
  echo "== variant [$v]"
  for i in 1 2; do timeout 120 python3 tools/rows7s_one.py 0 0 2>&1 | grep staged; done
done
rm -f cnmnet_amd/lib/conv_rows_staged.o

#!/bin/bash
# Round 4: what feeds conv_glds_kernel.  Rebuilds conv_mfma.hip with -DGLDS_PROBE=n (1: pixel pieces fetched for the first tap only -- the
# traffic of a halo-staged kernel; 2: filter pieces for the first k-step only; 3: both; results are wrong on purpose) and times
# every fp16 layer shape of a step (tools/f16_conv_probe.py).  The MFMAs, the fragment reads and the LDS writes of the DMA stay.
# 4: the pixel-piece DMA INSTRUCTIONS are not issued past the first tap (the issue count of a halo-staged kernel), 6 = 4 + 2.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
export F16_PROBE_NOCHECK=1   # modes 4 / 6 leave stale LDS contents in the product: the tile variants no longer agree
run() {
  local tag=$1; shift
  rm -f cnmnet_amd/lib/conv_mfma.o
  env "$@" python3 -m cnmnet_amd.build > /tmp/build.log 2>&1 || { tail -3 /tmp/build.log; return; }
  echo "== $tag"
  timeout 600 python3 tools/f16_conv_probe.py 2>&1 | cut -c1-60 | tail -27
}
run "shipped" X=1
run "pixel pieces: first tap only" CNM_EXTRA_HIPCC_FLAGS="-DGLDS_PROBE=1"
run "filter pieces: first k-step only" CNM_EXTRA_HIPCC_FLAGS="-DGLDS_PROBE=2"
run "neither" CNM_EXTRA_HIPCC_FLAGS="-DGLDS_PROBE=3"
run "pixel pieces: NO DMA instruction past the first tap" CNM_EXTRA_HIPCC_FLAGS="-DGLDS_PROBE=4"
run "that + filter pieces from zero-size descriptors" CNM_EXTRA_HIPCC_FLAGS="-DGLDS_PROBE=6"
rm -f cnmnet_amd/lib/conv_mfma.o
python3 -m cnmnet_amd.build > /tmp/build.log 2>&1

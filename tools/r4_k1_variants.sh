#!/bin/bash
# Round 4: does the plane sweep want more resident waves?  The kernel runs 4 waves per SIMD (one 1024-thread workgroup, 128 VGPRs);
# these builds trade registers (spills) and box size for 6 / 8 waves per SIMD with the existing switches (GPU box).
cd "$(dirname "$0")/.."
tools/k1_variants.sh "" \
  "-DSWEEP_CAP=1672 -DSWEEP_MINW=8" \
  "-DSWEEP_TH=12 -DSWEEP_CAP=1672 -DSWEEP_MINW=6" \
  "-DSWEEP_TH=8 -DSWEEP_CAP=1100 -DSWEEP_MINW=6" \
  "-DSWEEP_TH=8 -DSWEEP_CAP=800 -DSWEEP_MINW=8" \
  "-DSWEEP_TH=8 -DSWEEP_CAP=1672 -DSWEEP_MINW=4" \
  ""

#!/bin/bash
# One GPU-box pass over the plane-sweep kernel: variants, stress geometries, then the parity tests that use it.
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
{
tools/k1_variants.sh "-DSWEEP_STATS" "" "-DSWEEP_AHEAD=1" "-DSWEEP_AHEAD=3" \
  "-DSWEEP_CAP=1100 -DSWEEP_MINW=6 -DSWEEP_AHEAD=1" \
  "-DSWEEP_STATS@2 192 256 64 6 2" "-DSWEEP_STATS@2 192 256 64 25 2" "-DSWEEP_STATS@2 192 256 64 60 2" "-DSWEEP_STATS@2 480 640 96 1 1" "-DSWEEP_STATS@8 480 640 96 1 0" "-DSWEEP_STATS@1 100 130 20 1 2" "-DSWEEP_STATS@1 64 64 128 1 2"
} 2>&1 | grep -v "coredump\|core dump\|Failed to write\|c4 launches ok\|nchw launch ok" | tee gpurun_out/k1_round.txt
python -m pytest tests/test_gpu_parity.py -q -m gpu -k "planesweep or homography" -x 2>&1 | tail -15 | tee -a gpurun_out/k1_round.txt

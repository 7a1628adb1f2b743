#!/bin/bash
# One GPU-box pass over the plane-sweep kernel: variants, stress geometries, kernel durations from rocprofv3, then the
# parity tests that use it.
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
{
tools/k1_variants.sh "-DSWEEP_STATS" "" "-DSWEEP_AHEAD=2" \
  "-DSWEEP_STATS@2 192 256 64 6 2" "-DSWEEP_STATS@2 192 256 64 25 2" "-DSWEEP_STATS@2 192 256 64 60 2" "-DSWEEP_STATS@2 480 640 96 1 1" "@8 480 640 96 1 0" "-DSWEEP_STATS@1 100 130 20 1 2" "-DSWEEP_STATS@1 64 64 128 1 2" "-DSWEEP_STATS@1 64 64 8 1 2"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize tools/k1_bench.hip -o /tmp/k1_st 2>/dev/null
rm -rf /tmp/k1st; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/k1st -- /tmp/k1_st 8 192 256 64 stats 1 0 > /dev/null 2>&1
echo "rocprofv3 --kernel-trace --stats, 8 frames x 2 sources, 192x256x64:"; cat /tmp/k1st/*/*kernel_stats.csv | head -4
} 2>&1 | grep -v "coredump\|core dump\|Failed to write\|c4 launches ok\|nchw launch ok\|pure c4" | tee gpurun_out/k1_round.txt
python -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline_sizes.py -q -m gpu -k "planesweep or homography or getvolume or config4" -x 2>&1 | tail -5 | tee -a gpurun_out/k1_round.txt

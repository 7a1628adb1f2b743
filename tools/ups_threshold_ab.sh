#!/bin/bash
# [r6] which up_conv layers should run fused with their upsampling (cnm_tune_upsampled_min_pixels): bench steps of both engines per threshold.
# 196608 (shipped fp32): all four fused; 200000: depthNet's upconv2 (16 x 96 x 128 output pixels) unfused; 400000: + the refine net's two upconv1; 800000: none fused.
cd "$(dirname "$0")/.."
for rep in 1 2; do
for thr in 0 200000 400000 800000; do
  for prec in ${PRECS:-f16}; do
    if [ $thr = 0 ]; then unset CNM_TUNE; else export CNM_TUNE="upsampled_min_pixels=$thr"; fi
    timeout 300 python3 bench.py --precision $prec --steps 30 --warmup 5 --no-roofline --no-secondary --no-cpu-baseline --no-live-traffic 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('thr $thr $prec', round(d['value'],1), 'frames/s', round(d['ms_per_step'],3), 'ms')"
  done
done
done

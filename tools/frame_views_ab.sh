#!/bin/bash
# [r6] depthNet reading the caller's frame slices in place (CNM_FRAME_VIEWS=1) against four contiguous copies per call (0): bench step, fp32 and fp16; GPU box, alternating.
cd "$(dirname "$0")/.."
for rep in 1 2 3; do
  for v in 0 1; do
    for prec in f32 f16; do
      echo -n "frame_views=$v $prec: "
      CNM_FRAME_VIEWS=$v python3 bench.py --precision $prec --no-cpu-baseline --no-secondary --no-live-traffic --no-roofline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.1f frames/s  %.3f ms' % (d['value'], d['ms_per_step']))"
    done
  done
done

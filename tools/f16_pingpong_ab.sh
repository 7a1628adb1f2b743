#!/bin/bash
# Round 4: conv_glds_kernel with two wave groups one segment apart (-DGLDS_PINGPONG=1) against the lock-step schedule: the fp16 tests,
# every fp16 layer shape of a step (tools/f16_conv_probe.py) and a short fp16 bench, alternating builds on one box.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
run() {
  local tag=$1; shift
  rm -f cnmnet_amd/lib/conv_mfma.o
  env "$@" python3 -m cnmnet_amd.build > /tmp/build.log 2>&1 || { tail -3 /tmp/build.log; return; }
  echo "== $tag"
  timeout 900 python3 -m pytest tests -m gpu -x -q -k "f16 or fp16 or half or c8 or glds" 2>&1 | tail -2
  timeout 600 python3 tools/f16_conv_probe.py 2>&1 | cut -c1-60 | tail -27
  timeout 300 python3 bench.py --precision f16 --steps 30 --warmup 5 --no-roofline --no-secondary --no-cpu-baseline --no-live-traffic 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   bench f16', round(d['value'],1), 'frames/s', round(d['ms_per_step'],3), 'ms')"
}
run "ping-pong" CNM_EXTRA_HIPCC_FLAGS="-DGLDS_PINGPONG=1"
run "lock-step (shipped)" X=1
run "ping-pong" CNM_EXTRA_HIPCC_FLAGS="-DGLDS_PINGPONG=1"
rm -f cnmnet_amd/lib/conv_mfma.o
python3 -m cnmnet_amd.build > /tmp/build.log 2>&1

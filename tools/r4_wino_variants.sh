#!/bin/bash
# Round 4: build-time variants of the staged 36-point kernel, the sum line of tools/wino36s_probe.py time for each (GPU box).
# First pass (kept in DESIGN 4.2): static s_setprio on either wave half, channel-block-fastest unit order.  Second pass: register
# allocation -- tools/hotloop_proxy.sh shows the greedy allocator's class-priority switch removes every VGPR spill of the dominant
# instance (no scratch reloads in the phase loop).
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
P="-mllvm -greedy-regclass-priority-trumps-globalness=1"
for v in "" "$P" "$P -mllvm -disable-machine-licm" "" "$P"; do
  rm -f cnmnet_amd/lib/conv_winograd4s.o
  CNM_EXTRA_HIPCC_FLAGS="$v" python3 -m cnmnet_amd.build > /tmp/build.log 2>&1 || { tail -3 /tmp/build.log; continue; }
  echo "== variant [$v]"
  timeout 300 python3 tools/wino36s_probe.py time 2>&1 | tail -1
  timeout 300 python3 bench.py --steps 30 --warmup 5 --no-roofline --no-secondary --no-cpu-baseline --no-live-traffic 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   bench', round(d['value'],1), 'frames/s', round(d['ms_per_step'],3), 'ms')"
done
rm -f cnmnet_amd/lib/conv_winograd4s.o
python3 -m cnmnet_amd.build > /tmp/build.log 2>&1

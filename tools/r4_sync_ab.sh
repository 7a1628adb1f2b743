#!/bin/bash
# Round 4 A/B (GPU box): the staged kernels with the round-3 hand-off protocol (0 / 1 flags re-armed by the consumer, sources under
# tools/ab_r3/, not committed: `mkdir tools/ab_r3 && for f in conv_winograd4s conv_rows_staged; do git show f5a2a92:cnmnet_amd/csrc/$f.hip > tools/ab_r3/$f.hip; done`) against the generation protocol of this round, same box, alternating builds: sum line of the probe.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p /tmp/ab_new && cp cnmnet_amd/csrc/conv_winograd4s.hip cnmnet_amd/csrc/conv_rows_staged.hip /tmp/ab_new/
for v in new old new old; do
  if [ $v = old ]; then cp tools/ab_r3/*.hip cnmnet_amd/csrc/; else cp /tmp/ab_new/*.hip cnmnet_amd/csrc/; fi
  touch cnmnet_amd/csrc/conv_winograd4s.hip cnmnet_amd/csrc/conv_rows_staged.hip
  python3 -m cnmnet_amd.build > /tmp/build.log 2>&1 || { tail -3 /tmp/build.log; continue; }
  echo "== $v"
  timeout 300 python3 tools/wino36s_probe.py time 2>&1 | tail -1
  timeout 300 python3 bench.py --steps 30 --warmup 5 --no-roofline --no-secondary --no-cpu-baseline --no-live-traffic 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   bench', round(d['value'],1), 'frames/s', round(d['ms_per_step'],3), 'ms')"
done
cp /tmp/ab_new/*.hip cnmnet_amd/csrc/
python3 -m cnmnet_amd.build > /tmp/build.log 2>&1

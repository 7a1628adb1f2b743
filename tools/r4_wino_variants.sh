#!/bin/bash
# Round 4: build-time variants of the staged kernels (GPU box): the sum line of tools/wino36s_probe.py time and a short bench for each.
# First pass: static s_setprio on either wave half, channel-block-fastest unit order.  Second pass: register allocation
# -- tools/hotloop_proxy.sh: the greedy allocator's class-priority order + no machine LICM (now cnmnet_amd/build.py FILE_FLAGS).
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
P="-mllvm -greedy-regclass-priority-trumps-globalness=1 -mllvm -disable-machine-licm"
run() {   # run <tag> <env assignments...>
  local tag=$1; shift
  rm -f cnmnet_amd/lib/conv_winograd4s.o cnmnet_amd/lib/conv_rows_staged.o
  env "$@" python3 -m cnmnet_amd.build > /tmp/build.log 2>&1 || { tail -3 /tmp/build.log; return; }
  echo "== $tag"
  timeout 300 python3 tools/wino36s_probe.py time 2>&1 | tail -1
  timeout 300 python3 bench.py --steps 30 --warmup 5 --no-roofline --no-secondary --no-cpu-baseline --no-live-traffic 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   bench', round(d['value'],1), 'frames/s', round(d['ms_per_step'],3), 'ms')"
}
run "file flags (shipped)" X=1
run "no file flags" CNM_NO_FILE_FLAGS=1
run "file flags + the same on conv_rows_staged" CNM_EXTRA_HIPCC_FLAGS="$P"
run "file flags (shipped)" X=1
run "no file flags" CNM_NO_FILE_FLAGS=1
rm -f cnmnet_amd/lib/conv_winograd4s.o cnmnet_amd/lib/conv_rows_staged.o
python3 -m cnmnet_amd.build > /tmp/build.log 2>&1

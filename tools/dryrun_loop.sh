#!/bin/bash
# Root-causing the rare SIGABRT of the multi-rank dry runs, GPU box: the 8-rank train / eval dry runs of
# tests/test_gpu_bench_multi.py, N times each, ONE attempt per run, every run's stderr kept (faulthandler + C++ stacks on).
#   tools/dryrun_loop.sh <n_train> <n_eval> [outdir]
cd "$(dirname "$0")/.."
NT=${1:-10}; NE=${2:-10}; OUT=${3:-gpurun_out/dryrun}
mkdir -p $OUT
export PYTHONFAULTHANDLER=1 TORCH_SHOW_CPP_STACKTRACES=1 CNM_BENCH_BACKEND=gloo CNM_BENCH_DEVICE=0
unset RANK WORLD_SIZE LOCAL_RANK MASTER_ADDR MASTER_PORT
fail=0
run() {  # kind index cmd...
  kind=$1; i=$2; shift 2
  t0=$(date +%s)
  timeout 900 "$@" > $OUT/${kind}_$i.out 2> $OUT/${kind}_$i.err; rc=$?
  t1=$(date +%s)
  ok=$(grep -c '^{' $OUT/${kind}_$i.out)
  echo "$kind run $i: rc $rc, json lines $ok, $((t1-t0)) s" | tee -a $OUT/summary.txt
  if [ $rc -ne 0 ] || [ "$ok" != "1" ]; then
    fail=$((fail+1))
    echo "---- stderr of the failed run (filtered)" | tee -a $OUT/summary.txt
    grep -v "amdgpu.ids" $OUT/${kind}_$i.err | grep -i -B2 -A25 "abort\|signal\|terminate\|what()\|fatal\|Error\|error:" | head -150 | tee -a $OUT/summary.txt
  else
    rm -f $OUT/${kind}_$i.err $OUT/${kind}_$i.out
  fi
}
for i in $(seq 1 $NT); do run train $i python bench.py --mode train --gpus 8 --steps 2 --warmup 1 --samples-per-gpu 1; done
for i in $(seq 1 $NE); do run eval $i python bench.py --gpus 8 --steps 3 --warmup 1 --frames-per-gpu 1 --no-roofline --no-secondary; done
echo "dry runs: $NT train + $NE eval, failures $fail" | tee -a $OUT/summary.txt

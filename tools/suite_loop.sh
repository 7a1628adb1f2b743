#!/bin/bash
# [r6] the WHOLE GPU suite N times in a row on one box (the only context the multi-rank SIGABRT ever appeared in: DESIGN 6), one attempt per
# test, every multi-rank launch logged per rank (tests/test_gpu_bench_multi.py).  Summary -> gpurun_out/suite_loop_r6.txt (appended).
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
N=${1:-5}
box=$(hostname | tail -c 9)
mkdir -p gpurun_out
for i in $(seq 1 $N); do
  t0=$(date +%s)
  python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/suite_loop_r6_${box}_$i.log 2>&1
  rc=$?
  echo "box $box run $i: rc $rc, $(($(date +%s) - t0)) s, $(tail -1 gpurun_out/suite_loop_r6_${box}_$i.log)" | tee -a gpurun_out/suite_loop_r6.txt
  [ $rc -eq 0 ] && rm -f gpurun_out/suite_loop_r6_${box}_$i.log
done
echo "failed multi-rank launches kept: $(ls gpurun_out/multi_rank_logs 2>/dev/null | wc -l)" | tee -a gpurun_out/suite_loop_r6.txt
tail -n 40 gpurun_out/multi_rank_resources.txt 2>/dev/null | grep -c "rank 0" | sed 's/^/resource records (rank 0 lines in the tail): /'

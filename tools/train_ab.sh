#!/bin/bash
# [r6] the training step's two launch-count changes, A/B on one box, alternating: CNM_PACK_BATCH (the step's 3x3 filter packs as one launch)
# and CNM_BN_PARTIALS (BatchNorm in two launches per direction).  train_wo_normal as a HIP graph, B = 4, 192 x 256.
cd "$(dirname "$0")/.."
for rep in 1 2; do
  for cfg in "0 0" "1 0" "0 1" "1 1"; do
    set -- $cfg
    echo -n "pack_batch=$1 bn_partials=$2: "
    CNM_PACK_BATCH=$1 CNM_BN_PARTIALS=$2 timeout 300 python3 tools/train_bench.py 4 graph 2>/dev/null | tail -1
  done
done
echo -n "train (normals) graph, both on: "; timeout 300 python3 tools/train_bench.py 4 graph normals 2>/dev/null | tail -1
echo -n "train_wo_normal eager, both on: "; timeout 300 python3 tools/train_bench.py 4 2>/dev/null | tail -1

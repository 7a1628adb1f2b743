#!/bin/bash
# [r6] weight-gradient GEMMs: persistent stream-K launch against the split form (GPU box, alternating, train_wo_normal as a HIP graph, B = 4).
# share = how many ranges may share a tile on average for a launch to take the stream-K form (0 = every launch).
cd "$(dirname "$0")/.."
for rep in 1 2; do
  echo -n "split form everywhere: "; CNM_WGRAD_STREAMK=0 timeout 300 python3 tools/train_bench.py 4 graph 2>/dev/null | tail -1
  for sh in 0 1 2 3 4 8; do
    echo -n "stream-K, share $sh: "
    CNM_WGRAD_STREAMK=1 CNM_WGRAD_SK_SHARE=$sh timeout 300 python3 tools/train_bench.py 4 graph 2>/dev/null | tail -1
  done
done
for v in 0 1; do echo -n "train (normals) wgrad_streamk=$v (default share): "; CNM_WGRAD_STREAMK=$v timeout 300 python3 tools/train_bench.py 4 graph normals 2>/dev/null | tail -1; done

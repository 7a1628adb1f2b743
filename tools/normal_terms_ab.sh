#!/bin/bash
# [r6] the `train` step's surface-normal loss terms: fused kernels (CNM_FUSED_NORMAL_TERMS=1) against the torch expression (0); GPU box, alternating.
cd "$(dirname "$0")/.."
for rep in 1 2 3; do
  for v in 0 1; do echo -n "fused_normal_terms=$v: "; CNM_FUSED_NORMAL_TERMS=$v timeout 300 python3 tools/train_bench.py 4 graph normals 2>/dev/null | tail -1; done
done

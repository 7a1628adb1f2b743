#!/bin/bash
# [r6] weight-gradient GEMMs: scalar-offset loader of the Winograd-domain launches against the general coordinate walk (GPU box, alternating).
cd "$(dirname "$0")/.."
for rep in 1 2 3; do
  for v in 0 1; do
    echo -n "wgrad_linear=$v: "; CNM_WGRAD_LINEAR=$v timeout 300 python3 tools/train_bench.py 4 graph 2>/dev/null | tail -1
  done
done
echo -n "train (normals): "; timeout 300 python3 tools/train_bench.py 4 graph normals 2>/dev/null | tail -1

#!/bin/bash
# Round 5: the rebuilt plane sweep (24-byte column texels, two-sample loop, buffer stores) at 8 / 4 waves per SIMD (GPU box).
cd "$(dirname "$0")/.."
tools/k1_variants.sh "$@"

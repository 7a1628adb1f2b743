#!/bin/bash
# fp16 engine (GPU box): rocprofv3 kernel summary of `bench.py --precision f16` -> gpurun_out/f16_kernel_stats.csv, f16_bench.json
set -euo pipefail
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
export TMPDIR=/tmp
mkdir -p gpurun_out; rm -rf /tmp/f16st
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/f16st -- python3 bench.py --precision f16 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-live-traffic > gpurun_out/f16_bench.json 2>/dev/null
cp "$(find /tmp/f16st -name '*kernel_stats.csv' | head -1)" gpurun_out/f16_kernel_stats.csv
head -25 gpurun_out/f16_kernel_stats.csv | cut -c1-150; tail -c 600 gpurun_out/f16_bench.json

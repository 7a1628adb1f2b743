#!/bin/bash
set -euo pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r4
mkdir -p "$O"
env | grep -i -E "rocp|preload" || true
prof() {
  local tag=$1; shift
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats_$tag" -- python3 bench.py "$@" > "$O/prof_$tag.json" 2> "$O/prof_$tag.err"
  cp "$(find "$O/stats_$tag" -name '*kernel_stats.csv' | head -1)" "$O/bench_kernel_stats_$tag.csv"
  rm -rf "$O/stats_$tag"
}
prof default --no-cpu-baseline --no-secondary --no-live-traffic
prof serial --side-stream 0 --no-cpu-baseline --no-secondary --no-live-traffic
grep "conv_winograd36s_f32_kernel<16, false, 0, 4, false>" "$O"/bench_kernel_stats_*.csv | cut -c1-220
python3 -c "
import json
for t in ('default','serial'):
    d=json.loads(open('$O/prof_%s.json'%t).read().strip().splitlines()[-1]); print(t, d['roofline'].get('sclk_mhz_sustained'), d['roofline']['avg_launch_ms'])"

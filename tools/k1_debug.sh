#!/bin/bash
cd "$(dirname "$0")/.."
{
hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/bufload_probe.hip -o /tmp/bp 2>/dev/null
for m in 5 0 1 2 3 4; do timeout 60 /tmp/bp $m 2>&1 | grep -v "coredump\|core dump\|Failed to write" | tail -2; done
} 2>&1 | tee gpurun_out/k1_debug.txt

#!/bin/bash
# Host memory while a command runs (GPU box): MemAvailable sampled twice a second, the cgroup's limit and peak.  tools/mem_watch.sh <cmd...>
cd "$(dirname "$0")/.."
echo "MemTotal $(grep MemTotal /proc/meminfo | awk '{print int($2/1024)}') MB, available at start $(grep MemAvailable /proc/meminfo | awk '{print int($2/1024)}') MB; cgroup memory.max $(cat /sys/fs/cgroup/memory.max 2>/dev/null) current $(cat /sys/fs/cgroup/memory.current 2>/dev/null); cpus $(nproc)"
( while true; do grep MemAvailable /proc/meminfo | awk '{print int($2/1024)}'; sleep 0.5; done ) > /tmp/memwatch.txt &
W=$!
"$@"; rc=$?
kill $W 2>/dev/null
echo "exit code $rc; MemAvailable min $(sort -n /tmp/memwatch.txt | head -1) MB over $(wc -l < /tmp/memwatch.txt) samples; cgroup memory.peak $(cat /sys/fs/cgroup/memory.peak 2>/dev/null)"

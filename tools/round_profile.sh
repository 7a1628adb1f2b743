#!/bin/bash
# Round profile (GPU box): bench line, rocprofv3 kernel summaries (default and one-stream), PMC passes (HBM traffic,
# matrix-pipe utilisation, plane-sweep VALU counters), training-step kernel summary.  Output: gpurun_out/r5/.
set -euo pipefail
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
export TMPDIR=/tmp
O=gpurun_out/r5
rm -rf "$O"; mkdir -p "$O"
timeout 900 python3 bench.py > "$O/bench_line.json" 2> "$O/bench_line.err" || { tail -5 "$O/bench_line.err"; exit 1; }
prof() {   # prof <tag> <bench args...>: kernel summary of one bench command
  local tag=$1; shift
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats_$tag" -- python3 bench.py "$@" > "$O/prof_$tag.json" 2> "$O/prof_$tag.err"
  cp "$(find "$O/stats_$tag" -name '*kernel_stats.csv' | head -1)" "$O/bench_kernel_stats_$tag.csv"
  rm -rf "$O/stats_$tag"
}
prof default --no-cpu-baseline --no-secondary --no-live-traffic
prof serial --side-stream 0 --no-cpu-baseline --no-secondary --no-live-traffic
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$O/pmc_$C" -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-secondary --no-live-traffic > "$O/pmc_$C.log" 2>&1
done
python3 tools/pmc_traffic.py "$O/pmc_FETCH_SIZE" "$O/pmc_WRITE_SIZE" "$O/pmc_traffic.json" "$O/pmc_hbm_traffic.txt" > /dev/null
timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$O/pmc_mfma" -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-secondary --no-live-traffic > "$O/pmc_mfma.log" 2>&1
python3 tools/pmc_mfma.py "$O/pmc_mfma" "$O/pmc_mfma_util.txt" > /dev/null
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES --output-format csv -d "$O/pmc_valu" -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-secondary --no-live-traffic > "$O/pmc_valu.log" 2>&1
python3 tools/pmc_summary.py "$O/pmc_valu" | grep -i "planesweep" > "$O/pmc_planesweep_valu.txt" || true
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats_train" -- python3 tools/train_bench.py > "$O/train_bench.txt" 2> "$O/train_bench.err" || true
sed -i 's/$/   [under rocprofv3]/' "$O/train_bench.txt"
timeout 300 python3 tools/train_bench.py 4 2>/dev/null | tail -1 >> "$O/train_bench.txt" || true
timeout 300 python3 tools/train_bench.py 4 graph 2>/dev/null | tail -1 >> "$O/train_bench.txt" || true
timeout 600 python3 tools/wgrad_sweep.py > "$O/wgrad_sweep.txt" 2>/dev/null || true
timeout 600 python3 tools/f16_conv_probe.py > "$O/f16_conv_probe.txt" 2>/dev/null || true
timeout 300 python3 tools/f16_step_layers.py 2>/dev/null | grep -v amdgpu.ids > "$O/f16_step_layers.txt" || true
timeout 300 python3 tools/k1_context_probe.py 2>/dev/null | grep -v amdgpu.ids > "$O/k1_context_probe_box.txt" || true
bash tools/f16_profile.sh > /dev/null 2>&1 || true
# the plane sweep on its own: wave-cycle split (issuing / waiting / stalled), LDS counters, the sample loop in isolation, one traced launch
K1_FLAGS="" bash tools/k1_pmc.sh > "$O/k1_pmc.txt" 2>&1 || true
bash tools/k1_loop_probe.sh > "$O/k1_loop_probe.txt" 2>&1 || true
bash tools/r5_k1_variants.sh "-DSWEEP_TRACE" "" "-DSWEEP_NOSTORE" "@4 480 640 96" 2>&1 | grep "^\[\|first unit\|check" > "$O/k1_harness.txt" || true
python3 tools/k1_trace_report.py gpurun_out/k1_trace.txt > "$O/k1_trace.txt" 2>&1 || true
K1_SERIES=1 bash tools/k1_in_step.sh > "$O/k1_in_step.txt" 2>&1 || true
/tmp/wta > /dev/null 2>&1 || { hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/wino_tile_ablation.hip -o /tmp/wta 2>/dev/null && /tmp/wta > "$O/wino_tile_ablation.txt"; } || true
f=$(find "$O/stats_train" -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" "$O/train_kernel_stats.csv"
rm -rf "$O"/pmc_FETCH_SIZE "$O"/pmc_WRITE_SIZE "$O"/pmc_mfma "$O"/pmc_valu "$O"/stats_train
ls -la "$O"; head -c 1500 "$O/bench_line.json"; echo; head -6 "$O/bench_kernel_stats_serial.csv" | cut -c1-160; cat "$O/pmc_planesweep_valu.txt"

#!/bin/bash
# Round-6 profile (GPU box): bench line, rocprofv3 kernel summaries (default and one-stream), PMC passes (HBM traffic, matrix-pipe
# utilisation, plane-sweep VALU counters), config-4 kernel summary + plane-sweep PMC at config 4, training-step kernel summary.
# Output: gpurun_out/r6/ ; tools/copy_profiles6.sh copies the summaries into profiles/.
set -uo pipefail
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
export TMPDIR=/tmp
O=gpurun_out/r6
rm -rf "$O"; mkdir -p "$O"
timeout 1200 python3 bench.py > "$O/bench_line.json" 2> "$O/bench_line.err" || { tail -5 "$O/bench_line.err"; }
prof() {   # prof <tag> <bench args...>: kernel summary of one bench command
  local tag=$1; shift
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats_$tag" -- python3 bench.py "$@" > "$O/prof_$tag.json" 2> "$O/prof_$tag.err"
  cp "$(find "$O/stats_$tag" -name '*kernel_stats.csv' | head -1)" "$O/bench_kernel_stats_$tag.csv"
  rm -rf "$O/stats_$tag"
}
prof default --no-cpu-baseline --no-secondary --no-live-traffic
prof serial --side-stream 0 --no-cpu-baseline --no-secondary --no-live-traffic
prof graph --graph --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-live-traffic --no-roofline
prof f16 --precision f16 --side-stream 0 --no-cpu-baseline --no-secondary --no-live-traffic --no-roofline
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$O/pmc_$C" -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-secondary --no-live-traffic > "$O/pmc_$C.log" 2>&1
done
python3 tools/pmc_traffic.py "$O/pmc_FETCH_SIZE" "$O/pmc_WRITE_SIZE" "$O/pmc_traffic.json" "$O/pmc_hbm_traffic.txt" > /dev/null
timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$O/pmc_mfma" -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-secondary --no-live-traffic > "$O/pmc_mfma.log" 2>&1
python3 tools/pmc_mfma.py "$O/pmc_mfma" "$O/pmc_mfma_util.txt" > /dev/null
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES --output-format csv -d "$O/pmc_valu" -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-secondary --no-live-traffic > "$O/pmc_valu.log" 2>&1
python3 tools/pmc_summary.py "$O/pmc_valu" | grep -i "planesweep" > "$O/pmc_planesweep_valu.txt" || true
# config 4: kernel summary of the frame pipeline at 640 x 480 x 96, B = 4, S = 4
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats_c4" -- python3 tools/config4_run.py > "$O/config4_run.txt" 2>&1 || true
f=$(find "$O/stats_c4" -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" "$O/config4_kernel_stats.csv"; rm -rf "$O/stats_c4"
# the plane sweep at config 4 under the SQ counters (standalone harness, prebuilt tools/bin/k1_plain.bin), separate passes
for set in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA"; do
  tag=$(echo $set | cut -d' ' -f1)
  rm -rf "$O/k1c4_$tag"
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$O/k1c4_$tag" -- tools/bin/k1_plain.bin 8 480 640 96 pmc 1 0 > "$O/k1c4_$tag.log" 2>&1
  python3 tools/pmc_summary.py "$O/k1c4_$tag" | grep planesweep_kernel
  rm -rf "$O/k1c4_$tag"
done > "$O/k1_config4_pmc.txt" 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats_train" -- python3 tools/train_bench.py > "$O/train_bench.txt" 2> "$O/train_bench.err" || true
sed -i 's/$/   [under rocprofv3]/' "$O/train_bench.txt"
timeout 300 python3 tools/train_bench.py 4 2>/dev/null | tail -1 >> "$O/train_bench.txt" || true
timeout 300 python3 tools/train_bench.py 4 graph 2>/dev/null | tail -1 >> "$O/train_bench.txt" || true
f=$(find "$O/stats_train" -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" "$O/train_kernel_stats.csv"
timeout 300 python3 tools/f16_step_layers.py 2>/dev/null | grep -v amdgpu.ids > "$O/f16_step_layers.txt" || true
rm -rf "$O"/pmc_FETCH_SIZE "$O"/pmc_WRITE_SIZE "$O"/pmc_mfma "$O"/pmc_valu "$O"/stats_train
ls -la "$O"; tail -c 1500 "$O/bench_line.json"; echo; head -6 "$O/bench_kernel_stats_serial.csv" | cut -c1-160; cat "$O/pmc_planesweep_valu.txt"; cat "$O/train_bench.txt"

#!/bin/bash
# Copy the summaries of tools/round_profile.sh (gpurun_out/r5/) into profiles/ under their committed names.
set -euo pipefail
cd "$(dirname "$0")/.."
S=gpurun_out/r5; D=profiles
cp $S/bench_line.json $D/r5_bench_line.json
cp $S/bench_kernel_stats_default.csv $D/r5_bench_kernel_stats.csv
cp $S/bench_kernel_stats_serial.csv $D/r5_bench_kernel_stats_serial.csv
cp $S/pmc_hbm_traffic.txt $D/r5_pmc_hbm_traffic.txt
cp $S/pmc_mfma_util.txt $D/r5_pmc_mfma_util.txt
cp $S/pmc_planesweep_valu.txt $D/r5_pmc_planesweep_valu.txt
cp $S/pmc_traffic.json $D/r5_pmc_traffic.json
cp $S/train_bench.txt $D/r5_train_bench.txt
cp $S/train_kernel_stats.csv $D/r5_train_kernel_stats.csv
[ -f gpurun_out/f16_kernel_stats.csv ] && cp gpurun_out/f16_kernel_stats.csv $D/r5_f16_kernel_stats.csv
[ -f $S/config4_kernel_stats.csv ] && cp $S/config4_kernel_stats.csv $D/r5_config4_kernel_stats.csv
[ -f $S/wgrad_sweep.txt ] && cp $S/wgrad_sweep.txt $D/r5_wgrad_sweep.txt
[ -f $S/f16_conv_probe.txt ] && cp $S/f16_conv_probe.txt $D/r5_f16_conv_probe.txt
[ -s $S/f16_step_layers.txt ] && cp $S/f16_step_layers.txt $D/r5_f16_step_layers.txt
for f in k1_pmc k1_loop_probe k1_harness k1_trace k1_in_step wino_tile_ablation; do [ -s $S/$f.txt ] && cp $S/$f.txt $D/r5_$f.txt; done
git status --short $D

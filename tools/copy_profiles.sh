#!/bin/bash
# Copy the summaries of tools/round_profile.sh (gpurun_out/r4/) into profiles/ under their committed names.
set -euo pipefail
cd "$(dirname "$0")/.."
S=gpurun_out/r4; D=profiles
cp $S/bench_line.json $D/r4_bench_line.json
cp $S/bench_kernel_stats_default.csv $D/r4_bench_kernel_stats.csv
cp $S/bench_kernel_stats_serial.csv $D/r4_bench_kernel_stats_serial.csv
cp $S/pmc_hbm_traffic.txt $D/r4_pmc_hbm_traffic.txt
cp $S/pmc_mfma_util.txt $D/r4_pmc_mfma_util.txt
cp $S/pmc_planesweep_valu.txt $D/r4_pmc_planesweep_valu.txt
cp $S/pmc_traffic.json $D/r4_pmc_traffic.json
cp $S/train_bench.txt $D/r4_train_bench.txt
cp $S/train_kernel_stats.csv $D/r4_train_kernel_stats.csv
[ -f gpurun_out/f16_kernel_stats.csv ] && cp gpurun_out/f16_kernel_stats.csv $D/r4_f16_kernel_stats.csv
[ -f $S/config4_kernel_stats.csv ] && cp $S/config4_kernel_stats.csv $D/r4_config4_kernel_stats.csv
[ -f $S/wgrad_sweep.txt ] && cp $S/wgrad_sweep.txt $D/r4_wgrad_sweep.txt
[ -f $S/f16_conv_probe.txt ] && cp $S/f16_conv_probe.txt $D/r4_f16_conv_probe.txt
git status --short $D

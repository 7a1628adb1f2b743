#!/bin/bash
# Copy the summaries of tools/round6_profile.sh (gpurun_out/r6/) into profiles/ under their committed names.
set -uo pipefail
cd "$(dirname "$0")/.."
S=gpurun_out/r6; D=profiles
c() { [ -s "$S/$1" ] && cp "$S/$1" "$D/r6_$2"; }
c bench_line.json bench_line.json
c bench_kernel_stats_default.csv bench_kernel_stats.csv
c bench_kernel_stats_serial.csv bench_kernel_stats_serial.csv
c bench_kernel_stats_graph.csv bench_kernel_stats_graph.csv
c bench_kernel_stats_f16.csv f16_kernel_stats.csv
c pmc_hbm_traffic.txt pmc_hbm_traffic.txt
c pmc_mfma_util.txt pmc_mfma_util.txt
c pmc_planesweep_valu.txt pmc_planesweep_valu.txt
c pmc_traffic.json pmc_traffic.json
c config4_kernel_stats.csv config4_kernel_stats.csv
c config4_run.txt config4_run.txt
c k1_config4_pmc.txt k1_config4_pmc.txt
c train_bench.txt train_bench.txt
c train_kernel_stats.csv train_kernel_stats.csv
c f16_step_layers.txt f16_step_layers.txt
git status --short $D

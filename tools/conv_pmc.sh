#!/bin/bash
# Stall / issue counters of the convolution kernels inside bench.py (GPU box).
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/conv_pmc; rm -rf $O; mkdir -p $O
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU" "TA_TA_BUSY_sum TA_BUSY_avr TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_FLAT"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/p$i -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-secondary --no-live-traffic > $O/p$i.log 2>&1
  python3 tools/pmc_summary.py $O/p$i 2>/dev/null | grep -i "winograd36_f32_kernel<4, 3, false>\|rows_winograd_f32_kernel<7, 1\|conv_mfma_f32" | cut -c1-420
done 2>&1 | tee gpurun_out/conv_pmc.txt
rm -rf $O

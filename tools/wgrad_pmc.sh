#!/bin/bash
# [r6] PMC counters of the weight-gradient GEMM kernels inside the training step (GPU box); SQ / GRBM counters only, separate passes, kernel-trace only.
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVES" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC"; do
  tag=$(echo $set | cut -d' ' -f1)
  rm -rf /tmp/wgpmc_$tag
  CNM_WGRAD_STREAMK=${CNM_WGRAD_STREAMK:-1} timeout 400 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/wgpmc_$tag -- python3 tools/train_bench.py 4 > /tmp/wgpmc_$tag.log 2>&1
  echo "== $set"
  python3 tools/pmc_summary.py /tmp/wgpmc_$tag | grep -E "conv_wgrad|conv_winograd36s_f32_kernel<16, false, 0, 4, f" | cut -c1-400
done 2>&1 | tee gpurun_out/r6_wgrad_pmc.txt

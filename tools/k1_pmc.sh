#!/bin/bash
# PMC counters of the plane-sweep kernel in the standalone harness (GPU box); SQ / GRBM counters only, separate passes.
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize ${K1_FLAGS:-} tools/k1_bench.hip -o /tmp/k1_pmc 2>/dev/null || exit 1
mkdir -p gpurun_out
for set in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA"; do
  tag=$(echo $set | cut -d' ' -f1)
  rm -rf gpurun_out/k1pmc_$tag
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/k1pmc_$tag -- /tmp/k1_pmc 8 192 256 64 pmc 1 0 > gpurun_out/k1pmc_$tag.log 2>&1
  python3 tools/pmc_summary.py gpurun_out/k1pmc_$tag | grep planesweep_kernel
done 2>&1 | tee gpurun_out/k1_pmc.txt

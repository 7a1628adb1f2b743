#!/bin/bash
# Ablation timings + PMC of the LDS-staged Winograd kernel (GPU box; library built with CNM_EXTRA_HIPCC_FLAGS=-DWINO4S_ABLATE).
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
L="${1:-256 512 48 64 16}"
for r in 1 2; do for m in 0 4 32 64 3 35 67 7; do timeout 120 python3 tools/wino36s_one.py $L 1 $m 30 2>&1 | grep staged; done; done
O=gpurun_out/w36s_pmc; rm -rf $O; mkdir -p $O
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_VMEM SQ_INSTS_WAVE32_LDS SQ_INSTS_SMEM"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/p$i -- python3 tools/wino36s_one.py $L 1 0 10 > $O/p$i.log 2>&1
  python3 tools/pmc_summary.py $O/p$i 2>/dev/null | grep -i "winograd36s" | cut -c1-460
  grep -i "error\|invalid" $O/p$i.log | head -3
done
rm -rf $O

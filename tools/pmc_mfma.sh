# MFMA pipe utilisation per kernel from PMC counters (GPU box).  Always under `timeout`; SQ / GRBM counters only.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 400 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_mfma -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/pmc_mfma.log 2>&1
echo "rc=$?"
python3 tools/pmc_summary.py gpurun_out/pmc_mfma | sort -t= -k2 -n -r | head -30

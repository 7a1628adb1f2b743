# HBM traffic per kernel launch from PMC counters, one counter per pass (GPU box).  Always under `timeout`.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 400 rocprofv3 --kernel-trace --pmc $C --output-format csv -d gpurun_out/pmc_$C -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/pmc_$C.log 2>&1
  echo "pass $C rc=$?"
done
python3 tools/pmc_traffic.py gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE gpurun_out/pmc_traffic.json gpurun_out/pmc_hbm_traffic.txt

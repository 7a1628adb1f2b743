#!/bin/bash
# Full GPU pass (GPU box): parity tests, smoke, bench line.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python -m pytest tests -q -m gpu -x 2>&1 | tail -15 | tee gpurun_out/gpu_tests.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee gpurun_out/smoke.txt
python bench.py --steps 20 --warmup 5 > gpurun_out/bench_now.json 2> gpurun_out/bench_now.err; tail -c 3000 gpurun_out/bench_now.json

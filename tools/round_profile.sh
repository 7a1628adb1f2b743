# Refresh profiles/: default bench line + rocprofv3 kernel stats of the same command (GPU box).
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 400 python3 bench.py > gpurun_out/bench_line.json 2> gpurun_out/bench_line.err
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/stats -- python3 bench.py > gpurun_out/bench_prof.json 2> gpurun_out/bench_prof.err
f=$(find gpurun_out/stats -name '*kernel_stats.csv' | head -1)
cp "$f" gpurun_out/kernel_stats.csv
head -12 gpurun_out/kernel_stats.csv | cut -c1-150

# Like-for-like per-kernel profile: the refine decoders kept on one stream so that kernel durations do not overlap.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/stats_serial -- python3 bench.py --side-stream 0 > gpurun_out/bench_prof_serial.json 2> gpurun_out/bench_prof_serial.err
f=$(find gpurun_out/stats_serial -name '*kernel_stats.csv' | head -1)
cp "$f" gpurun_out/kernel_stats_serial.csv
head -4 gpurun_out/kernel_stats_serial.csv | cut -c1-150
tail -c 1500 gpurun_out/bench_prof_serial.json | head -c 700

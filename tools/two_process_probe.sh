#!/bin/bash
# Round 4: two independent single-rank processes of bench.py on ONE GPU at the same time, eager launches against hipGraph replays
# (are graph replays of two processes on one device pathological by themselves?  DESIGN 6)
export TMPDIR=/tmp
mkdir -p gpurun_out/r4p
run2() { # args...
  python3 bench.py "$@" > gpurun_out/r4p/a.json 2>/dev/null &
  p1=$!
  python3 bench.py "$@" > gpurun_out/r4p/b.json 2>/dev/null &
  p2=$!
  wait $p1 $p2
  for f in a b; do python3 -c "
import json,sys; d=json.loads(open('gpurun_out/r4p/$f.json').read().strip().splitlines()[-1]); print('   ', round(d['value'],1), d['unit'], round(d['ms_per_step'],2), 'ms', d.get('step_ms'))"; done
}
echo "== train eager x2"; run2 --mode train --steps 30 --warmup 3
echo "== train graph x2 (long enough to overlap)"; run2 --mode train --steps 60 --warmup 3 --graph
echo "== eval eager x2"; run2 --steps 200 --warmup 3 --no-roofline --no-secondary --no-cpu-baseline --no-live-traffic
echo "== eval graph x2"; run2 --steps 200 --warmup 3 --no-roofline --no-secondary --no-cpu-baseline --no-live-traffic --graph

#!/bin/bash
# VERDICT r4 item 7 (GPU box): the exposed part of the gradient exchange in the 2-rank dry run (both ranks on GPU 0, gloo: the
# exchange goes through the host and says nothing about xGMI -- the RELATIVE figures show what each cut hides): eager (backward
# hooks), two graphs (refine | depthNet), three graphs (refine | decoder | encoder).
cd "$(dirname "$0")/.."
export CNM_BENCH_BACKEND=gloo CNM_BENCH_DEVICE=0
unset RANK WORLD_SIZE LOCAL_RANK MASTER_ADDR MASTER_PORT
run() { "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); a=d['allreduce']; print('ms/step %.1f | buckets %d, launched before the end of backward %d, late %d | exposed %.1f ms | %s' % (d['ms_per_step'], a['buckets'], a['launched_from_backward_hooks'], a['launched_late'], a['exposed_ms'], d['config']['launch']))"; }
echo -n "eager:        "; run python bench.py --mode train --gpus 2 --steps 4 --warmup 2 --samples-per-gpu 2
echo -n "two graphs:   "; CNM_GRAPH_CUTS=1 run python bench.py --mode train --gpus 2 --steps 4 --warmup 2 --samples-per-gpu 2 --graph
echo -n "three graphs: "; CNM_GRAPH_CUTS=2 run python bench.py --mode train --gpus 2 --steps 4 --warmup 2 --samples-per-gpu 2 --graph

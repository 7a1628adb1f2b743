for rep in 1 2; do
for pol in plain nt; do
CNM_SWEEP_STORE=$pol python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline_planesweep']
print('$pol', round(d['value'],1), 'fps  K1 in-step median us', round(r['launch_ms']['median']*1e3,1), 'avg', round(r['avg_launch_ms']*1e3,1), 'frac', round(r['frac'],3), r['store_policy']['in_force'])"
done
done
python - <<'EOF'
import torch, ctypes
from cnmnet_amd import ops, _lib
lib=_lib.load()
med=(ctypes.c_float*2)()
for i in range(3):
    print('calibration', ops.calibrate_sweep_store('cuda', force=True), [round(x,1) for x in (lib.cnm_tune_sweep_store(99, ctypes.cast(med, ctypes.c_void_p)), med[0], med[1])])
EOF

cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline > gpurun_out/trace.log 2>&1
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/trace/**/*kernel_trace.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# take the last step: find last planesweep_kernel
idx=[i for i,r in enumerate(rows) if r['Kernel_Name'].startswith('void planesweep_kernel')]
a=idx[-1]; 
out=open('gpurun_out/trace_last_step.txt','w')
for r in rows[a-5:]:
    d=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3
    out.write("%9.1f us  grid %s wg %s  %s\n"%(d, r['Grid_Size_X'] if 'Grid_Size_X' in r else r.get('Grid_Size'), r.get('Workgroup_Size_X',r.get('Workgroup_Size')), r['Kernel_Name'][:70]))
out.close()
PY

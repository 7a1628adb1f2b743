#!/bin/bash
# rocprofv3 kernel trace of the plane-sweep harness (GPU box): per-launch duration and the gaps between back-to-back launches.
#   tools/k1_rocprof.sh "<-D flags>"
cd "$(dirname "$0")/.."
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize $1 tools/k1_bench.hip -o /tmp/k1_prof 2>/dev/null || { echo build failed; exit 1; }
rm -rf /tmp/k1prof_out; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/k1prof_out -- /tmp/k1_prof 8 192 256 64 prof 1 0 x > /tmp/k1prof.log 2>&1
f=$(find /tmp/k1prof_out -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, statistics as st
rows = [r for r in csv.DictReader(open(sys.argv[1])) if 'planesweep_kernel<1,' in r['Kernel_Name'] or 'planesweep_kernelILi1' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
dur = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in rows]
gap = [(int(b['Start_Timestamp']) - int(a['End_Timestamp'])) / 1e3 for a, b in zip(rows, rows[1:])]
gap = [g for g in gap if g < 50]
print(f'{len(rows)} launches: duration median {st.median(dur):.1f} min {min(dur):.1f} max {max(dur):.1f} us | gap to the next launch (back to back) median {st.median(gap):.1f} us | vgpr {rows[0].get("VGPR_Count")} sgpr {rows[0].get("SGPR_Count")} lds {rows[0].get("LDS_Block_Size")} scratch {rows[0].get("Scratch_Size")}')
PY

#!/bin/bash
# Per-kernel register / spill / LDS table of one source file (hipcc -Rpass-analysis=kernel-resource-usage).
# usage: tools/kernel_resources.sh cnmnet_amd/csrc/conv_winograd4s.hip [extra hipcc flags]
src=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c "$src" -o /tmp/kr_$$.o -Rpass-analysis=kernel-resource-usage "$@" 2>&1 |
python3 -c '
import sys,re
cur=None; rows=[]
for l in sys.stdin:
    m=re.search(r"Function Name: (\S+)",l)
    if m: cur={"name":m.group(1)}; rows.append(cur); continue
    m=re.search(r"remark:\s+([A-Za-z /\[\]]+): (\d+)",l)
    if m and cur is not None: cur[m.group(1).strip()]=int(m.group(2))
import subprocess
for r in rows:
    n=subprocess.run(["c++filt",r["name"]],capture_output=True,text=True).stdout.strip()
    n=re.sub(r"\(.*","",n)[:90]
    print("%-90s vgpr %3s agpr %3s spill %3s sgpr %3s sspill %3s scratch %4s lds %6s occ %s"%(n,r.get("VGPRs"),r.get("AGPRs"),r.get("VGPRs Spill"),r.get("TotalSGPRs"),r.get("SGPRs Spill"),r.get("ScratchSize [bytes/lane]"),r.get("LDS Size [bytes/block]"),r.get("Occupancy [waves/SIMD]")))
'
rm -f /tmp/kr_$$.o

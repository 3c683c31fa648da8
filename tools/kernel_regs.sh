#!/bin/bash
# usage: regs.sh file.hip [extra flags] -> kernel, VGPRs, AGPRs, spills, scratch, LDS
f=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -Rpass-analysis=kernel-resource-usage -c $f -o /tmp/regs/x.o --cuda-device-only "$@" 2>&1 | python3 -c "
import sys,re
cur=None; rows=[]
for line in sys.stdin:
    m=re.search(r'Function Name: (\S+)',line)
    if m: cur={'name':m.group(1)}; rows.append(cur); continue
    for key in ('VGPRs','AGPRs','ScratchSize \[bytes/lane\]','VGPR Spill','LDS Size \[bytes/block\]','Occupancy \[waves/SIMD\]'):
        m=re.search(r' '+key+r': (\d+)',line)
        if m and cur is not None: cur[key]=m.group(1)
import subprocess
for r in rows:
    n=subprocess.run(['c++filt',r['name']],capture_output=True,text=True).stdout.strip()
    print(n[:90].ljust(90), 'V',r.get('VGPRs'),'A',r.get('AGPRs'),'spill',r.get('VGPR Spill'),'scr',r.get('ScratchSize \[bytes/lane\]'),'lds',r.get('LDS Size \[bytes/block\]'),'occ',r.get('Occupancy \[waves/SIMD\]'))
"

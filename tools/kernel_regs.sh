#!/bin/bash
# Register / scratch usage per kernel of one HIP source:  tools/kernel_regs.sh attention.hip [filter]
cd "$(dirname "$0")/../omnibiote_amd/csrc" || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 $EXTRA -Rpass-analysis=kernel-resource-usage -c "$1" -o /tmp/kregs_probe.o 2>&1 | python3 -c "
import sys,re
flt=sys.argv[1] if len(sys.argv)>1 else ''
cur=None;rows={}
for l in sys.stdin:
    m=re.search(r'Function Name: (\S+)',l)
    if m: cur=m.group(1); rows[cur]={}
    for k,n in (('VGPRs','vgpr'),('AGPRs','agpr'),('ScratchSize \[bytes/lane\]','scratch'),('Occupancy \[waves/SIMD\]','occ'),('TotalSGPRs','sgpr')):
        m=re.search(r'    '+k+r': (\d+)',l)
        if m and cur: rows[cur][n]=int(m.group(1))
import subprocess
for k,v in rows.items():
    name=subprocess.run(['c++filt',k],capture_output=True,text=True).stdout.strip()
    name=re.sub(r'\(anonymous namespace\)::','',name); name=re.sub(r'\(.*','',name)
    if flt in name: print(f'{name:70s}', v)
" "$2"

#!/bin/bash
# usage: tools_resusage.sh file.hip  -> kernel name, VGPRs, AGPRs, scratch, occupancy, LDS
hipcc --offload-arch=gfx950 -mcode-object-version=5 -O3 -std=c++17 -fPIC -c -Rpass-analysis=kernel-resource-usage -o /dev/null "$@" 2>&1 | python3 -c "
import sys,re
cur=None; rows=[]
for l in sys.stdin:
    m=re.search(r'remark: (?:\s*)(Function Name|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]|SGPRs): (\S+)', l)
    if not m: continue
    k,v=m.groups()
    if k=='Function Name': cur={'name':v}; rows.append(cur)
    elif cur is not None: cur[k.split(' ')[0]]=v
for r in rows: print(r)
"

#!/bin/bash
# usage: tools/resource_usage.sh csrc-file.hip [extra hipcc flags]  ->  kernel | sgpr vgpr scratch occupancy lds   (one line per kernel)
f=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -Wall -Wno-unused-function "$@" -c -Rpass-analysis=kernel-resource-usage $f -o /dev/null 2>&1 | python3 -c "
import sys,re
cur={}
for l in sys.stdin:
    m=re.search(r'Function Name: (\S+)',l)
    if m: cur={'n':m.group(1)}
    for k,pat in (('s','TotalSGPRs: (\d+)'),('v',' VGPRs: (\d+)'),('sc','ScratchSize \[bytes/lane\]: (\d+)'),('o','Occupancy \[waves/SIMD\]: (\d+)'),('l','LDS Size \[bytes/block\]: (\d+)')):
        m=re.search(pat,l)
        if m: cur[k]=m.group(1)
    if 'l' in cur:
        print(cur['n'][:90].ljust(90), 'sgpr',cur.get('s'),'vgpr',cur.get('v'),'scratch',cur.get('sc'),'occ',cur.get('o'),'lds',cur.get('l')); cur={}
"

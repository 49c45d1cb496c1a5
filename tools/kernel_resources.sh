#!/bin/bash
# Register / scratch / LDS usage of every kernel of one HIP source (compile-time report).
#   tools/kernel_resources.sh go-curdleproofs_amd/csrc/msm_reduce_kernels.hip
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics \
  -Rpass-analysis=kernel-resource-usage -c "$1" -o /dev/null 2>&1 | python3 -c "
import sys,re
for l in sys.stdin:
    m=re.search(r'Function Name: (\S+)',l)
    if m: print(); print(m.group(1)[:70],end=' ')
    for k in [' VGPRs',' AGPRs','VGPRs Spill','SGPRs Spill','ScratchSize','Occupancy','LDS Size']:
        m=re.search(r'remark: +'+k.strip()+r'[^:]*: *(\d+)',l) if k.strip() not in ('VGPRs','AGPRs') else re.search(r'remark: +'+k.strip()+r': *(\d+)',l)
        if m: print(k.strip().replace(' ','')+'='+m.group(1),end=' ')
print()
"

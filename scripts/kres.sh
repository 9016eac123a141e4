#!/bin/bash
# scripts/kres.sh <file.hip> [filter]: per-kernel register / scratch / occupancy table from hipcc's resource remarks
f=$1; pat=${2:-.}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c "$f" -o /tmp/kres.$$.o -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c "
import sys,re
rows=[];cur=None
for l in sys.stdin:
    m=re.search(r'remark:\s+(.*?) \[-Rpass', l)
    if not m: continue
    t=m.group(1).strip()
    if t.startswith('Function Name'):
        cur={'fn':t.split(': ',1)[1]}; rows.append(cur)
    elif cur is not None and ':' in t:
        k,v=t.rsplit(':',1); cur[k.strip()]=v.strip()
import subprocess
for r in rows:
    name=subprocess.run(['c++filt',r['fn']],capture_output=True,text=True).stdout.strip()
    name=re.sub(r'uf::\(anonymous namespace\)::','',name); name=re.sub(r'\(.*','',name).replace('void ','')
    if not re.search(sys.argv[1],name): continue
    print(f\"{name[:70]:70s} VGPR {r.get('VGPRs','?'):>4s} AGPR {r.get('AGPRs','?'):>4s} SGPR {r.get('TotalSGPRs','?'):>4s} scratch {r.get('ScratchSize [bytes/lane]','?'):>5s} occ {r.get('Occupancy [waves/SIMD]','?')}\")
" "$pat"
rm -f /tmp/kres.$$.o

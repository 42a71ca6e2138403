#!/bin/bash
# Kernel resource summary of one .hip file: kernels with scratch, and the top VGPR users.
# usage: tools/kres.sh semantic-segmentation_amd/csrc/conv_tile.hip
f=$1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c "$f" -o /tmp/kres.o -Rpass-analysis=kernel-resource-usage 2>&1 |
python3 -c '
import sys,re
cur=None; rows=[]
for l in sys.stdin:
    m=re.search(r"Function Name: (\S+)",l)
    if m: cur={"name":m.group(1)}; rows.append(cur); continue
    for k in ("VGPRs","AGPRs","ScratchSize \[bytes/lane\]","Occupancy \[waves/SIMD\]","LDS Size \[bytes/block\]"):
        m=re.search(r"\s"+k+r": (\d+)",l)
        if m and cur is not None: cur[k.split(" ")[0]]=int(m.group(1))
print("%d kernels"%len(rows))
bad=[r for r in rows if r.get("ScratchSize",0)>0]
print("with scratch:",[(r["name"][:90],r["ScratchSize"]) for r in bad])
for r in sorted(rows,key=lambda r:-r.get("VGPRs",0))[:5]: print(r.get("VGPRs"),r.get("AGPRs"),r.get("Occupancy"),r["name"][:100])
'

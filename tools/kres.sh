#!/bin/bash
# Compact per-kernel resource table (VGPRs, spills, occupancy, LDS) of one .hip file: tools/kres.sh rk_kernels_split.hip [extra flags]
cd "$(dirname "$0")/../rakau_amd/csrc"
f=$1; shift
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off -fno-slp-vectorize -fvisibility=hidden "$@" \
  -Rpass-analysis=kernel-resource-usage -c "$f" -o /tmp/kres.o 2>&1 | python3 -c '
import sys,re,subprocess
cur={}
rows=[]
for l in sys.stdin:
    m=re.search(r"remark: (.*?): (.*?) \[-Rpass", l)
    if not m:
        m=re.search(r"remark: (Function Name): (\S+)", l)
        if not m: continue
    k,v=m.group(1).strip(),m.group(2).strip()
    if k=="Function Name":
        if cur: rows.append(cur)
        cur={"name":v}
    else: cur[k]=v
if cur: rows.append(cur)
for r in rows:
    try: name=subprocess.run(["c++filt",r["name"]],capture_output=True,text=True).stdout.strip()
    except Exception: name=r["name"]
    name=re.sub(r"\(.*","",name).replace("void rk::","")
    print("%-44s vgpr %4s agpr %3s sgpr %4s spillV %3s spillS %3s occ %2s lds %6s scratch %s"%(name[:44],r.get("VGPRs"),r.get("AGPRs"),r.get("TotalSGPRs"),r.get("VGPR Spill"),r.get("SGPR Spill"),r.get("Occupancy [waves/SIMD]"),r.get("LDS Size [bytes/block]"),r.get("ScratchSize [bytes/lane]")))
'

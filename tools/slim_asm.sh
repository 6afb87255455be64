#!/bin/bash
# Device assembly of the fp32 / Q = 0 / bh instantiations only (-DRK_SLIM; seconds instead of minutes):
#   tools/slim_asm.sh rk_kernels_list.hip /tmp/out.s [-DX=..]...
cd "$(dirname "$0")/../rakau_amd/csrc" || exit 1
f=$1; o=$2; shift 2
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off -fno-slp-vectorize -fvisibility=hidden -DRK_SLIM "$@" \
  --offload-device-only -S "$f" -o "$o" -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c '
import sys,re,subprocess
cur={}; rows=[]
for l in sys.stdin:
    m=re.search(r"remark: (.*?): (.*?) \[-Rpass", l)
    if not m:
        m=re.search(r"remark: (Function Name): (\S+)", l)
        if not m:
            if "error" in l: print(l, end="")
            continue
    k,v=m.group(1).strip(),m.group(2).strip()
    if k=="Function Name":
        if cur: rows.append(cur)
        cur={"name":v}
    else: cur[k]=v
if cur: rows.append(cur)
for r in rows:
    name=subprocess.run(["c++filt",r["name"]],capture_output=True,text=True).stdout.strip()
    name=re.sub(r"\(.*","",name).replace("void rk::","")
    print("%-44s vgpr %4s sgpr %4s occ %2s lds %6s scratch %s"%(name[:44],r.get("VGPRs"),r.get("TotalSGPRs"),r.get("Occupancy [waves/SIMD]"),r.get("LDS Size [bytes/block]"),r.get("ScratchSize [bytes/lane]")))
'

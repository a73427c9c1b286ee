#!/bin/bash
# Print VGPR/AGPR/spill/LDS/occupancy per kernel for the gfx950 build (hipcc -Rpass-analysis).
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Rpass-analysis=kernel-resource-usage \
  -o /tmp/_vame_res.so "${@:-vame_amd/csrc/gru_seq.hip vame_amd/csrc/gemm.hip vame_amd/csrc/elementwise.hip}" 2>&1 |
python3 -c '
import sys,re
cur=None
rows=[]
for line in sys.stdin:
    m=re.search(r"Function Name: (\S+)",line)
    if m: cur={"name":m.group(1)}; rows.append(cur); continue
    m=re.search(r"remark:\s+([A-Za-z][^:]*?): (\S+)",line)
    if m and cur is not None: cur[m.group(1).strip()]=m.group(2)
    elif "error" in line: print(line.rstrip())
for r in rows:
    print("%-70s vgpr=%s agpr=%s spill=%s scratch=%s lds=%s occ=%s"%(r["name"][:70],r.get("VGPRs"),r.get("AGPRs"),r.get("VGPRs Spill"),r.get("ScratchSize [bytes/lane]"),r.get("LDS Size [bytes/block]"),r.get("Occupancy [waves/SIMD]")))
'

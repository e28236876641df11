#!/bin/bash
# Where a kernel translation unit touches scratch memory (spills, stack-passed arguments), by source line:  tools/dbg/scratch_sites.sh k_scatter_layered_staged_tri
TU=$1
D=$(mktemp -d); cd $D
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -fno-gpu-flush-denormals-to-zero -Wno-unused-result -Wno-unused-value -gline-tables-only -x hip -c /root/repo/shimmer_amd/csrc/$TU.hip -I /root/repo/shimmer_amd/csrc -o k.o -save-temps 2>/dev/null
python3 - <<PY
import re
from collections import Counter, defaultdict
asm=open('$TU-hip-amdgcn-amd-amdhsa-gfx950.s').read().splitlines()
files={}
for l in asm:
    m=re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?',l)
    if m: files[int(m.group(1))]=(m.group(3) or m.group(2)).split('/')[-1]
fn=None; loc=None; sc=defaultdict(Counter)
for l in asm:
    m=re.match(r'^(_Z\S+):',l)
    if m: fn=m.group(1)[:70]
    m=re.match(r'\s*\.loc\s+(\d+)\s+(\d+)',l)
    if m: loc=(files.get(int(m.group(1)),'?'),int(m.group(2)))
    t=l.strip()
    if t.startswith('scratch_'): sc[fn][(t.split()[0],loc)]+=1
for f,c in sc.items():
    print(f, sum(c.values()))
    for k,n in sorted(c.items(), key=lambda kv: kv[0][1] or ('',0)): print('   ',k,n)
for l in asm:
    if re.search(r'\.(name|vgpr_count|vgpr_spill_count|private_segment_fixed_size|sgpr_spill_count):',l): print(l.strip())
PY
rm -rf $D

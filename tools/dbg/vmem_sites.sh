#!/bin/bash
# Static census of the vector-memory instructions of a kernel translation unit by source file (and the top lines):  tools/dbg/vmem_sites.sh k_vertex_tex [function-substring]
TU=$1; FN=${2:-}
D=$(mktemp -d); cd $D
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -fno-gpu-flush-denormals-to-zero -Wno-unused-result -Wno-unused-value -gline-tables-only -x hip -c /root/repo/shimmer_amd/csrc/$TU.hip -I /root/repo/shimmer_amd/csrc -o k.o -save-temps 2>/dev/null
python3 - "$FN" <<PY
import re, sys
from collections import Counter, defaultdict
want = sys.argv[1]
asm=open('$TU-hip-amdgcn-amd-amdhsa-gfx950.s').read().splitlines()
files={}
for l in asm:
    m=re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?',l)
    if m: files[int(m.group(1))]=(m.group(3) or m.group(2)).split('/')[-1]
fn=None; loc=None; by=defaultdict(Counter); ops=defaultdict(Counter)
for l in asm:
    m=re.match(r'^(_Z\S+):',l)
    if m: fn=m.group(1)
    m=re.match(r'\s*\.loc\s+(\d+)\s+(\d+)',l)
    if m: loc=(files.get(int(m.group(1)),'?'),int(m.group(2)))
    t=l.strip()
    if t.startswith(('global_load','flat_load','buffer_load','scratch_load','global_store','flat_store','scratch_store','global_atomic','flat_atomic')) and fn and want in fn:
        by[fn][loc]+=1; ops[fn][t.split()[0]]+=1
for f,c in by.items():
    print(f[:90], sum(c.values()), dict(ops[f]))
    perfile=Counter()
    for (fl,ln),n in c.items(): perfile[fl]+=n
    print('   by file:', dict(perfile))
    for k,n in c.most_common(25): print('     ',k,n)
PY
rm -rf $D

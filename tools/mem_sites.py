#!/usr/bin/env python3
"""Where a kernel translation unit touches memory, by source line, from the compiler's own assembly: every scratch access (register spills, stack-passed
arguments, objects whose address escaped) and, with --all, every vector-memory instruction. Compiles shimmer_amd/csrc/<tu>.hip as the Makefile does plus
-gline-tables-only -save-temps. This is how round 4 found that the 1 240 B of scratch of k_vertex<textured> were not spilled registers but the kernel's SceneView
(a stack object because the texture evaluators took it by reference), and the `wi` argument of the LayeredBxDF interface calls.
    python tools/mem_sites.py k_vertex_tex [--fn SUBSTRING] [--all] [--top N]
"""
import argparse
import re
import subprocess
import tempfile
from collections import Counter, defaultdict
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
CSRC = ROOT / "shimmer_amd" / "csrc"
FLAGS = "--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt " \
        "-fno-gpu-flush-denormals-to-zero -Wno-unused-result -Wno-unused-value -gline-tables-only".split()
VMEM = ("global_load", "flat_load", "buffer_load", "scratch_load", "global_store", "flat_store", "buffer_store", "scratch_store", "global_atomic", "flat_atomic")


def main():
    ap = argparse.ArgumentParser(description=__doc__.splitlines()[0])
    ap.add_argument("tu", help="translation unit under shimmer_amd/csrc, without .hip")
    ap.add_argument("--fn", default="", help="only functions whose mangled name contains this")
    ap.add_argument("--all", action="store_true", help="every vector-memory instruction, not only scratch")
    ap.add_argument("--top", type=int, default=40, help="lines listed per function")
    args = ap.parse_args()
    with tempfile.TemporaryDirectory() as tmp:
        subprocess.check_call(["/opt/rocm/bin/hipcc", *FLAGS, "-x", "hip", "-c", str(CSRC / f"{args.tu}.hip"), "-I", str(CSRC), "-o", f"{tmp}/k.o", "-save-temps"],
                              cwd=tmp, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        asm = Path(tmp, f"{args.tu}-hip-amdgcn-amd-amdhsa-gfx950.s").read_text().splitlines()
    files = {}
    for l in asm:
        m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
        if m:
            files[int(m.group(1))] = (m.group(3) or m.group(2)).split("/")[-1]
    fn, loc = None, None
    sites, ops = defaultdict(Counter), defaultdict(Counter)
    for l in asm:
        m = re.match(r"^(_Z\S+):", l)
        if m:
            fn = m.group(1)
        m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", l)
        if m:
            loc = (files.get(int(m.group(1)), "?"), int(m.group(2)))
        t = l.strip()
        if fn and args.fn in fn and t.startswith(VMEM if args.all else ("scratch_",)):
            sites[fn][(t.split()[0], loc)] += 1
            ops[fn][t.split()[0]] += 1
    for f, c in sites.items():
        name = subprocess.run(["c++filt", f], capture_output=True, text=True).stdout.strip().replace("(anonymous namespace)::", "")
        print(f"{re.sub(r'[(].*', '', name)[:100]}: {sum(c.values())} instructions {dict(ops[f])}")
        per_file = Counter()
        for (op, lc), n in c.items():
            per_file[lc[0] if lc else "?"] += n
        print("   by file:", dict(per_file))
        for (op, lc), n in sorted(c.items(), key=lambda kv: -kv[1])[:args.top]:
            print(f"      {n:4d}  {op:24s} {lc[0]}:{lc[1]}" if lc else f"      {n:4d}  {op}")
    for l in asm:
        if re.search(r"\.(name|vgpr_count|vgpr_spill_count|sgpr_spill_count|private_segment_fixed_size):", l):
            print(l.strip())


if __name__ == "__main__":
    main()

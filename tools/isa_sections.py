#!/usr/bin/env python3
"""Per-section instruction census of the traversal kernels from the compiler's own assembly ("nobody has shown the ISA").

Compiles shimmer_amd/csrc/k_trace.hip exactly as the Makefile does plus -gline-tables-only -save-temps, and attributes every instruction of
k_trace5<closest, GEN> / k_trace5<any, GEN> to a section of trace5_body by the source line its .loc names (inlined leaf functions —
shm/shapes.h, shm/fp.h — count towards the section that calls them: the leaf phase). Classes: VALU (v_*), SALU (s_* except waitcnt / nop /
branches), BRANCH (s_cbranch*, s_branch), VMEM (global_* / buffer_* / scratch_* / flat_*), LDS (ds_*), WAIT (s_waitcnt, s_nop).
The counts are STATIC (instructions in the code object per section); one loop iteration executes the refill test, at most one pop, one node
step, one leaf-mark / push, and — when enough lanes wait — the leaf phase, so per-iteration dynamic counts are the sums of the sections on
that path, not of all of them.
    python tools/isa_sections.py [-o out.txt]
"""
import argparse
import re
import subprocess
import sys
import tempfile
from collections import defaultdict
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
CSRC = ROOT / "shimmer_amd" / "csrc"
FLAGS = "--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt " \
        "-fno-gpu-flush-denormals-to-zero -Wno-unused-result -Wno-unused-value -gline-tables-only".split()


MARKS3 = [("setup", "template <bool ANY, bool TRI_ONLY, int LDS_N>"), ("refill", "// ---- refill idle lanes"),
          ("node_step (load + slab test)", "// ---- one uniform node step"), ("leaf-mark / push", "// the three outcomes as selects"),
          ("leaf_phase", "// ---- postponed leaf phase"), ("pop", "// ---- pop: a lane that missed"),
          ("retire", "// ---- retire finished rays"), ("epilogue", "unsigned long long w_prims = c_prims;"), ("end", "#define K3_PARAMS")]
MARKS5 = [("setup", "template <bool ANY, bool GEN, int LDS_N>"), ("refill (+ root test)", "// ---- refill idle lanes"),
          ("pair_step (2 x (load + slab test) + push)", "// ---- one uniform step: every lane that stands"),
          ("leaf_phase", "// ---- postponed leaf phase: lanes standing"), ("other_phase (GEN: sphere / patch / instance)", "// ---- GEN: the parked non-triangle tests"),
          ("pop", "// ---- pop: a lane whose two children"),
          ("retire", "// ---- retire finished rays"), ("epilogue", "unsigned long long wn = c_nodes;"), ("end", "#ifndef K5_CLOSEST_WAVES")]  # (MARKS3: the retired one-node-step body; profiles/r03_k_trace3_isa*.txt were made with it)


def sections_of(src_lines, marks, first_line=0):
    """Line ranges of a traversal body's sections, found by the comments that open them (so the census follows the source as it changes)."""
    out, pos = [], first_line
    for name, needle in marks:
        for i in range(pos, len(src_lines)):
            if needle in src_lines[i]:
                out.append((name, i + 1))
                pos = i + 1
                break
        else:
            raise SystemExit(f"section marker not found: {needle}")
    return [(out[k][0], out[k][1], out[k + 1][1] - 1) for k in range(len(out) - 1)]


def classify(op):
    if op.startswith(("s_cbranch", "s_branch")):
        return "BRANCH"
    if op.startswith(("s_waitcnt", "s_nop")):
        return "WAIT"
    if op.startswith("s_"):
        return "SALU"
    if op.startswith("ds_"):
        return "LDS"
    if op.startswith(("global_", "buffer_", "scratch_", "flat_")):
        return "VMEM"
    if op.startswith("v_"):
        return "VALU"
    return "OTHER"


def main():
    ap = argparse.ArgumentParser(description=__doc__.splitlines()[0])
    ap.add_argument("-o", "--out", help="also write the census to this file")
    args = ap.parse_args()  # (flags are parsed before anything is written: `--help` once became an output path)
    src = (CSRC / "k_trace.hip").read_text().splitlines()
    secs5 = sections_of(src, MARKS5)
    with tempfile.TemporaryDirectory() as tmp:
        subprocess.check_call(["/opt/rocm/bin/hipcc", *FLAGS, "-x", "hip", "-c", str(CSRC / "k_trace.hip"), "-I", str(CSRC), "-o", f"{tmp}/k.o", "-save-temps"],
                              cwd=tmp, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        asm = Path(tmp, "k_trace-hip-amdgcn-amd-amdhsa-gfx950.s").read_text().splitlines()
    files = {}
    for l in asm:
        m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
        if m:
            files[int(m.group(1))] = m.group(3) or m.group(2)
    lines_out = [f"# static instruction census of the traversal kernels by section of trace5_body / trace3_body (tools/isa_sections.py; hipcc {' '.join(FLAGS[:3])} ...)"]
    for label, want, secs in (("k_trace5<closest> (both-children step)", "k_trace5ILb0ELb0EE", secs5), ("k_trace5<any>", "k_trace5ILb1ELb0EE", secs5),
                              ("k_trace5<closest, GEN> (scenes with spheres / patches / instances)", "k_trace5ILb0ELb1EE", secs5), ("k_trace5<any, GEN>", "k_trace5ILb1ELb1EE", secs5)):
        start = next(i for i, l in enumerate(asm) if re.match(r"^_ZN.*" + want + r".*:", l))
        end = next(i for i in range(start, len(asm)) if asm[i].strip().startswith(".Lfunc_end"))
        counts = defaultdict(lambda: defaultdict(int))
        ops = defaultdict(lambda: defaultdict(int))
        cur = "setup"
        for l in asm[start:end]:
            m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", l)
            if m:
                fname, line = files.get(int(m.group(1)), ""), int(m.group(2))
                if fname.endswith("k_trace.hip"):
                    cur = next((n for n, a, b in secs if a <= line <= b), cur)
                continue  # (instructions of inlined header functions keep the section of the call site)
            t = l.strip()
            if not t or t.startswith((";", ".", "_Z")) or t.endswith(":"):
                continue
            op = t.split()[0]
            counts[cur][classify(op)] += 1
            ops[cur][op] += 1
        res = "\n".join(l for l in asm[end:end + 80] if re.search(r"\.(vgpr_count|sgpr_count|vgpr_spill_count|sgpr_spill_count|private_segment_fixed_size|group_segment_fixed_size)", l))
        lines_out.append(f"\n== {label} ==")
        lines_out.append(f"{'section':34s} {'VALU':>5s} {'SALU':>5s} {'BRANCH':>6s} {'VMEM':>5s} {'LDS':>4s} {'WAIT':>5s}   source lines")
        tot = defaultdict(int)
        for name, a, b in secs:
            c = counts.get(name, {})
            for k, v in c.items():
                tot[k] += v
            lines_out.append(f"{name:34s} {c.get('VALU', 0):5d} {c.get('SALU', 0):5d} {c.get('BRANCH', 0):6d} {c.get('VMEM', 0):5d} {c.get('LDS', 0):4d} {c.get('WAIT', 0):5d}   k_trace.hip:{a}-{b}")
        lines_out.append(f"{'total':34s} {tot['VALU']:5d} {tot['SALU']:5d} {tot['BRANCH']:6d} {tot['VMEM']:5d} {tot['LDS']:4d} {tot['WAIT']:5d}")
        for name in [n for n, _, _ in secs if n not in ("setup", "epilogue", "retire")]:
            top = sorted(ops.get(name, {}).items(), key=lambda kv: -kv[1])[:14]
            lines_out.append(f"  {name}: " + ", ".join(f"{k} x{v}" for k, v in top))
    text = "\n".join(lines_out) + "\n"
    if args.out:
        Path(args.out).write_text(text)
    print(text)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Registers, spills, LDS and scratch of every kernel in libshimmer_hip.so, read from the code objects' own metadata (the AMDGPU notes of the
ELF images embedded in the library) — the numbers occupancy follows from: waves per SIMD = min(8, 512 / VGPRs rounded up to 8, ...). (rocprofv3's
kernel-trace `vgpr` column is NOT this count: it read 40 for a kernel that holds 76.)   python tools/kernel_resources.py [library]"""
import re, subprocess, sys, tempfile, os
lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "shimmer_amd", "csrc", "libshimmer_hip.so")
data = open(lib, "rb").read()
readelf = "/opt/rocm/lib/llvm/bin/llvm-readelf"
rows, seen = [], set()
with tempfile.TemporaryDirectory() as tmp:
    for k, m in enumerate(re.finditer(b"\x7fELF\x02\x01\x01", data)):
        o = m.start()
        if int.from_bytes(data[o + 18:o + 20], "little") != 224:  # EM_AMDGPU
            continue
        shoff = int.from_bytes(data[o + 40:o + 48], "little")
        size = shoff + int.from_bytes(data[o + 58:o + 60], "little") * int.from_bytes(data[o + 60:o + 62], "little")
        fn = os.path.join(tmp, f"co{k}.elf")
        open(fn, "wb").write(data[o:o + size])
        cur = {}
        for line in subprocess.run([readelf, "--notes", fn], capture_output=True, text=True).stdout.splitlines():
            mm = re.match(r"\s*-?\s*\.(\w+):\s*(.*)", line)
            if not mm:
                continue
            key, val = mm.group(1), mm.group(2).strip()
            if key == "agpr_count" and cur:  # first key of a kernel's record
                rows.append(cur)
                cur = {}
            cur[key] = val
        if cur:
            rows.append(cur)
def demangle(n):
    try:
        d = subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip()
    except FileNotFoundError:
        d = n
    return re.sub(r"\(.*", "", d.replace("(anonymous namespace)::", "").replace("void ", ""))
print(f"{'kernel':44s} {'vgpr':>5} {'sgpr':>5} {'spill v/s':>10} {'lds B':>7} {'scratch B':>9} {'waves/SIMD (VGPR, LDS@256 thr)':>30}")
for r in rows:
    if "vgpr_count" not in r or r.get("name") in seen or not r.get("name", "").startswith("_Z"):
        continue
    seen.add(r["name"])
    v, lds = int(r["vgpr_count"]), int(r.get("group_segment_fixed_size", 0))
    by_v = min(8, 512 // max(8, (v + 7) // 8 * 8))
    by_l = min(8, 163840 // lds) if lds else 8
    print(f"{demangle(r['name'])[:44]:44s} {v:5d} {int(r['sgpr_count']):5d} {r.get('vgpr_spill_count', '?'):>5}/{r.get('sgpr_spill_count', '?'):<4} {lds:7d} {int(r.get('private_segment_fixed_size', 0)):9d} {by_v:>14d}, {by_l:d}")

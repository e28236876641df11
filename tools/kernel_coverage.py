#!/usr/bin/env python3
"""Every __global__ symbol of libshimmer_hip.so (from the code objects embedded in the library) against the kernels a rocprofv3 --kernel-trace --stats run launched.
    python tools/kernel_coverage.py <rocprofv3 output directory> [library]
Prints one line per kernel symbol: calls in the traced run (0 = the suite never reached this instantiation), then the totals."""
import csv, glob, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
trace_dir = sys.argv[1]
lib = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "shimmer_amd", "csrc", "libshimmer_hip.so")


def norm(name):
    """`void (anonymous namespace)::k_shade<false, true>(wf::ShadeArgs)` and `k_shade<false, true>` -> the same key."""
    n = name.strip().strip('"')
    n = re.sub(r"\s*\[clone [^\]]*\]", "", n)
    n = n.replace("(anonymous namespace)::", "")
    n = re.sub(r"^void\s+", "", n)
    depth, out = 0, []
    for ch in n:  # cut the argument list: the first "(" outside template brackets
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            break
        out.append(ch)
    return "".join(out).replace(" ", "").replace(".kd", "")


def library_kernels(path):
    data = open(path, "rb").read()
    readelf = "/opt/rocm/lib/llvm/bin/llvm-readelf"
    names = set()
    with tempfile.TemporaryDirectory() as tmp:
        for k, m in enumerate(re.finditer(b"\x7fELF\x02\x01\x01", data)):
            o = m.start()
            if int.from_bytes(data[o + 18:o + 20], "little") != 224:  # EM_AMDGPU
                continue
            shoff = int.from_bytes(data[o + 40:o + 48], "little")
            size = shoff + int.from_bytes(data[o + 58:o + 60], "little") * int.from_bytes(data[o + 60:o + 62], "little")
            fn = os.path.join(tmp, f"co{k}.elf")
            open(fn, "wb").write(data[o:o + size])
            for line in subprocess.run([readelf, "--notes", fn], capture_output=True, text=True).stdout.splitlines():
                mm = re.match(r"\s*-?\s*\.name:\s*(\S+)", line)
                if mm and mm.group(1).startswith("_Z") and not mm.group(1).endswith(".kd"):
                    names.add(mm.group(1))
    dem = subprocess.run(["c++filt"], input="\n".join(sorted(names)), capture_output=True, text=True).stdout.splitlines()
    return sorted({norm(d) for d in dem})


def traced_calls(d):
    """{kernel: launches} of a rocprofv3 --kernel-trace run: its rocpd SQLite database (ROCm 7: the default output; the `top_kernels` view), else the CSV forms."""
    calls = {}
    files = glob.glob(os.path.join(d, "**", "*.db"), recursive=True)
    if files:
        import sqlite3
        for f in files:
            for name, n in sqlite3.connect(f).cursor().execute("select name, total_calls from top_kernels"):
                k = norm(name)
                calls[k] = calls.get(k, 0) + int(n)
        return calls, files
    files = glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)
    if files:
        for f in files:
            for row in csv.DictReader(open(f)):
                k = norm(row.get("Name") or row.get("KernelName") or "")
                calls[k] = calls.get(k, 0) + int(row.get("Calls") or row.get("calls") or 0)
        return calls, files
    files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    for f in files:
        for row in csv.DictReader(open(f)):
            k = norm(row.get("Kernel_Name") or row.get("Name") or "")
            calls[k] = calls.get(k, 0) + 1
    return calls, files


kernels = library_kernels(lib)
calls, files = traced_calls(trace_dir)
print(f"# library {os.path.relpath(lib, ROOT)}: {len(kernels)} kernel symbols, {os.path.getsize(lib) / 1e6:.1f} MB; trace files: {len(files)} ({sum(calls.values())} launches of {len(calls)} distinct kernels)")
unreached = [k for k in kernels if calls.get(k, 0) == 0]
for k in kernels:
    print(f"{calls.get(k, 0):10d}  {k}")
foreign = sorted(k for k in calls if k not in set(kernels))
print(f"# kernels launched by the run that are not this library's (runtime / RCCL / test helpers): {len(foreign)}")
for k in foreign:
    print(f"#   {calls[k]:8d}  {k}")
print(f"# UNREACHED by the traced run: {len(unreached)} of {len(kernels)}")
for k in unreached:
    print(f"#   {k}")

#!/usr/bin/env python3
"""Joins gather_fetch's own dispatch list with rocprofv3's per-dispatch counters: counter value per requested byte / per record."""
import csv, glob, os, re, sys
out = sys.argv[1]
rows = []
for line in open(os.path.join(out, "plain.txt")):
    m = re.match(r"dispatch (\d+) pattern (\w+) array_MB (\d+) pass (\d+) requested_bytes (\d+) records (\d+) ms ([\d.]+) requested_GBs ([\d.]+)", line)
    if m:
        rows.append({"i": int(m.group(1)), "pattern": m.group(2), "mb": int(m.group(3)), "pass": int(m.group(4)), "bytes": int(m.group(5)),
                     "records": int(m.group(6)), "ms": float(m.group(7)), "gbs": float(m.group(8))})
counters = {}
for f in glob.glob(os.path.join(out, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    per = {}
    for r in csv.DictReader(open(f)):
        if not r["Kernel_Name"].startswith("k_"):
            continue
        per.setdefault(r["Counter_Name"], []).append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    for c, v in per.items():
        acc = {}
        for d, x in v:
            acc[d] = acc.get(d, 0.0) + x
        counters[c] = [acc[d] for d in sorted(acc)]
print("# tools/ubench/gather_fetch.hip under rocprofv3 --pmc (one dispatch per line; the untimed runs under the profiler are slower than `ms`, which is the plain run's)")
print(f"{'pattern':9s} {'MB':>5s} {'pass':>4s} {'ms':>8s} {'req GB/s':>9s} | " + " ".join(f"{c:>22s}" for c in sorted(counters)) + " | FETCH_SIZE KiB->bytes / requested   FETCH bytes per record")
for r in rows:
    vals = {c: (counters[c][r["i"]] if r["i"] < len(counters[c]) else float("nan")) for c in counters}
    fs = vals.get("FETCH_SIZE", float("nan")) * 1024.0
    print(f"{r['pattern']:9s} {r['mb']:5d} {r['pass']:4d} {r['ms']:8.3f} {r['gbs']:9.1f} | " + " ".join(f"{vals[c]:22.0f}" for c in sorted(counters)) +
          f" | {fs / r['bytes']:8.3f} {fs / r['records']:12.1f}")

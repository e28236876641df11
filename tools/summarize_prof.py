#!/usr/bin/env python3
"""Summarise a tools/profile_gpu.sh output directory (rocprofv3 rocpd .db files): per-kernel time from the kernel-trace
pass and per-kernel PMC sums from each counter pass.  FETCH_SIZE on gfx950 reports half the bytes of a wide coalesced
read (MI355X_MICROARCH.md §HBM): raw and x2 figures are both printed; the access pattern here is 16-B gathers, for which
the guide says the absolute value is uncalibrated, so ratios between variants are what to read."""
import glob
import os
import re
import sqlite3
import sys
from collections import defaultdict

out = sys.argv[1]


def short(name):
    m = re.search(r"k_(trace\d?|shade\w*|vertex|scatter\w*|generate|film|expand_tiles|next_bounce|reset_heads3|fold\w*|emit_jobs|split_plain)", name)
    if not m:
        return name[:48]
    k = m.group(0)
    if k.startswith("k_scatter") or k.startswith("k_vertex") or k == "k_shade":  # template arguments name the class / scene class
        t = re.search(r"<([^>]*)>", name)
        if t:
            k += "<" + t.group(1).replace("true", "1").replace("false", "0").replace(" ", "") + ">"
    if k.startswith("k_trace"):
        t = re.search(r"k_trace\d?<(\w+)(?:, (\w+))?(?:, (\w+))?>", name)
        if t:
            k += "<any>" if t.group(1) == "true" else "<closest>"
            if k.startswith("k_trace5"):
                k += "+gen" if t.group(2) == "true" else ""   # k_trace5<ANY, GEN, HEAVY>
                k += "+heavy" if t.group(3) == "true" else ""
            else:
                k += "+sph" if t.group(2) == "false" else ""  # (profiles of rounds 1-4: k_trace3<ANY, TRI_ONLY>)
    return k


print(f"# {out}")
print("== kernel-trace --stats ==")
for f in glob.glob(os.path.join(out, "stats", "*.db")):
    cur = sqlite3.connect(f).cursor()
    for name, calls, total, avg, pct in cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels"):
        print(f"{short(name):24s} calls={calls:5d} total_ms={total/1e3:10.3f} avg_us={avg:10.2f} pct={pct:6.2f}")
    r = cur.execute("select name, vgpr_count, accum_vgpr_count, sgpr_count, lds_size, workgroup_x, grid_x from kernels group by name").fetchall()
    for name, v, a, s, lds, wg, grid in r:
        if "k_" in name:
            print(f"   {short(name):22s} vgpr={v} agpr={a} sgpr={s} lds={lds} wg={wg} grid={grid}")
traffic = {}
print("== PMC (sum over dispatches, per kernel) ==")
for d in sorted(glob.glob(os.path.join(out, "pmc_*"))):
    for f in glob.glob(os.path.join(d, "*.db")):
        cur = sqlite3.connect(f).cursor()
        acc = defaultdict(lambda: defaultdict(float))
        cnt = defaultdict(set)
        dur = defaultdict(float)
        for name, counter, value, disp, duration in cur.execute("select kernel_name, counter_name, value, dispatch_id, duration from counters_collection"):
            k = short(name)
            acc[k][counter] += value
            if disp not in cnt[k]:
                dur[k] += duration
            cnt[k].add(disp)
        for k in sorted(acc):
            if not k.startswith("k_"):
                continue
            for c, v in sorted(acc[k].items()):
                n = len(cnt[k])
                extra = ""
                if c == "FETCH_SIZE":
                    extra = f"  = {v*1024/1e9:8.3f} GB raw, {2*v*1024/1e9:8.3f} GB x2; {v*1024/n/1e6:8.2f} MB/dispatch raw; {v*1024/(dur[k]):7.1f} GB/s raw over profiled time"
                elif c == "WRITE_SIZE":
                    extra = f"  = {v*1024/1e9:8.3f} GB; {v*1024/n/1e6:8.2f} MB/dispatch"
                print(f"{k:24s} {c:22s} {v:20.1f}  ({n} dispatches, {dur[k]/1e6:9.3f} ms){extra}")
                if c in ("FETCH_SIZE", "WRITE_SIZE"):
                    traffic.setdefault(k, {})[c + "_bytes_per_dispatch_raw"] = v * 1024 / n
                    traffic[k]["dispatches"] = n
                elif c in ("SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU", "SQ_INSTS_VALU", "GRBM_GUI_ACTIVE", "SQ_WAIT_ANY", "SQ_WAVE_CYCLES", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_SCA", "SQ_INSTS_SALU"):
                    traffic.setdefault(k, {})[c] = v / n  # per dispatch: what bench.py's committed-profile fallback prices the VALU-issue roofline with (committed_pmc)
            # mean active lanes per VALU instruction (SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU) when that pass was collected
            if "SQ_THREAD_CYCLES_VALU" in acc[k] and acc[k].get("SQ_ACTIVE_INST_VALU", 0) > 0:
                lanes = acc[k]["SQ_THREAD_CYCLES_VALU"] / acc[k]["SQ_ACTIVE_INST_VALU"]
                traffic.setdefault(k, {})["valu_lanes_active"] = lanes
                print(f"{k:24s} {'lanes per VALU instr':22s} {lanes:20.1f}")
import json
json.dump(traffic, open(os.path.join(out, "traffic.json"), "w"), indent=1)

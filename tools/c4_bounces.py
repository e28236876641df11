#!/usr/bin/env python3
"""Per-bounce queue sizes and traversal launch times of the crown-proxy (C4, maxdepth 32): SHM_DEBUG=1 prints them from inside the library.
    python tools/c4_bounces.py [--json]    (--json: one JSON object — queue sizes, mean path length — for bench.py's C4 side result)"""
import json, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = "import sys; sys.path.insert(0, %r)\nfrom shimmer_amd import abi, scenes, render\nlib = abi.load_library()\nsc = scenes.crown_proxy(lib, 1000, 1400)\nr = render.Renderer(lib, sc.desc, 0)\np = render.make_params(seed=0, spp=256, max_depth=32)\nr.clear(); r.render_device(p)\n" % ROOT
out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, SHM_DEBUG="1"), capture_output=True, text=True).stderr
rows, cl, an, sh = [], [], [], []
for line in out.splitlines():
    if "bounce" in line and "traced" in line:
        rows.append(line.split("[shm] ")[1])
    elif "closest launch" in line: cl.append(float(line.split()[-2]))
    elif "any launch" in line: an.append(float(line.split()[-2]))
    elif "shade launch" in line: sh.append(float(line.split()[-2]))
if "--json" in sys.argv[1:]:
    q = [tuple(int(x) for x in re.match(r"bounce \d+: traced (\d+), next (\d+), shadow (\d+)", r).groups()) for r in rows]
    paths = q[0][0] if q else 0
    print(json.dumps({"paths": paths, "extension_queue_per_bounce": [a for a, _, _ in q], "shadow_queue_per_bounce": [c for _, _, c in q],
                      # path length in segments (camera ray + every extension ray) and in scattering vertices (bounces survived)
                      "mean_path_length_segments": sum(a for a, _, _ in q) / max(1, paths), "mean_bounces_survived": sum(b for _, b, _ in q) / max(1, paths),
                      "queue_occupancy_vs_bounce0": [round(a / max(1, paths), 5) for a, _, _ in q],
                      "bounces_with_half_the_paths": next((i for i, (a, _, _) in enumerate(q) if a * 2 < paths), None)}))
    sys.exit(0)
for i, row in enumerate(rows):
    print(row, "| closest %.2f ms any %.2f ms shade %.2f ms" % (cl[i] if i < len(cl) else 0, an[i] if i < len(an) else 0, sh[i] if i < len(sh) else 0))
print("totals: closest %.1f any %.1f shade %.1f ms" % (sum(cl), sum(an), sum(sh)))

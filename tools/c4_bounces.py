#!/usr/bin/env python3
"""Per-bounce queue sizes and traversal launch times of the crown-proxy (C4, maxdepth 32): SHM_DEBUG=1 prints them from inside the library."""
import os, subprocess, sys
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
for i, row in enumerate(rows):
    print(row, "| closest %.2f ms any %.2f ms shade %.2f ms" % (cl[i] if i < len(cl) else 0, an[i] if i < len(an) else 0, sh[i] if i < len(sh) else 0))
print("totals: closest %.1f any %.1f shade %.1f ms" % (sum(cl), sum(an), sum(sh)))

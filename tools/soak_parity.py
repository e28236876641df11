#!/usr/bin/env python3
"""Soak run of the GPU-vs-oracle parity check over many seeded random scenes (scenes.random_scene: every shape, material and light kind,
thin lens, textures for seed % 16 >= 12): film sums and visit counters must be bit-equal, scene after scene. The test suite runs seeds 0..15;
this runs any range.   python tools/soak_parity.py [first] [count] [quirks_off]   (quirks_off = 1: ShmRenderParams::disable_reference_quirks, the PBRT-v4 forms)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import oracle_py
from shimmer_amd import abi, render, scenes
first = int(sys.argv[1]) if len(sys.argv) > 1 else 16
count = int(sys.argv[2]) if len(sys.argv) > 2 else 100
quirks_off = len(sys.argv) > 3 and sys.argv[3] == "1"
lib = abi.load_library()
bad = []
t0 = time.time()
for seed in range(first, first + count):
    try:
        sc = scenes.random_scene(lib, seed)
    except Exception as e:
        print(f"seed {seed}: scene generation failed: {e}", flush=True)
        continue
    p = render.make_params(seed=100 + seed, spp=6, max_depth=7, regularize=bool(seed % 4 == 3), reference_quirks=not quirks_off)
    gpu, orc = render.Renderer(lib, sc.desc, 0), oracle_py.Oracle(sc.desc)
    fg, sg = gpu.render(p)
    fo, so = orc.render(p, n_threads=os.cpu_count() or 1)
    same = np.array_equal(fg.view(np.uint64), fo.view(np.uint64)) and all(sg[k] == so[k] for k in ("paths", "rays_closest", "rays_any", "nodes_closest", "tris_closest", "nodes_any", "tris_any"))
    if not same:
        bad.append(seed)
        print(f"seed {seed}: MISMATCH pixels {int((fg['rgb_sum'].view(np.uint64) != fo['rgb_sum'].view(np.uint64)).any(axis=-1).sum())} stats {[(k, sg[k], so[k]) for k in ('rays_closest','rays_any','nodes_closest') if sg[k] != so[k]]}", flush=True)
    gpu.close(); orc.close()
print(f"seeds {first}..{first + count - 1}{' (reference quirks OFF)' if quirks_off else ''}: {count - len(bad)} bit-equal, mismatches {bad}; nonfinite-aware compare; {time.time() - t0:.0f} s", flush=True)

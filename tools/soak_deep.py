#!/usr/bin/env python3
"""Soak of the deep-path machinery (late-bounce overlap, the fused tail launch, the specular / rough scatter split) against the CPU oracle: seeded variants of
the crown proxy (dispersive glass, rough gold, diffuse floor) and of the glass Cornell box at maxdepth 12..32; films and visit counters must be bit-equal.
    python tools/soak_deep.py [first] [count]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import oracle_py
from shimmer_amd import abi, render, scenes
first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 40
lib = abi.load_library()
bad = []
t0 = time.time()
for seed in range(first, first + count):
    rng = np.random.Generator(np.random.PCG64(seed))
    if seed % 4 == 3:
        sc = scenes.cornell_box(lib, 40 + int(rng.integers(0, 24)), 40 + int(rng.integers(0, 24)), glass=True)
    else:
        sc = scenes.crown_proxy(lib, 40 + int(rng.integers(0, 30)), 56 + int(rng.integers(0, 30)), level=int(rng.integers(1, 3)), n_glass=int(rng.integers(4, 20)),
                                n_gold=int(rng.integers(0, 8)), seed=1000 + seed)
    p = render.make_params(seed=500 + seed, spp=int(rng.integers(2, 9)), max_depth=int(rng.integers(12, 33)), regularize=bool(seed % 5 == 4))
    gpu, orc = render.Renderer(lib, sc.desc, 0), oracle_py.Oracle(sc.desc)
    fg, sg = gpu.render(p)
    fo, so = orc.render(p, n_threads=os.cpu_count() or 1)
    same = np.array_equal(fg.view(np.uint64), fo.view(np.uint64)) and all(sg[k] == so[k] for k in ("paths", "rays_closest", "rays_any", "nodes_closest", "tris_closest", "nodes_any", "tris_any"))
    if not same:
        bad.append(seed)
        print(f"seed {seed}: MISMATCH {[(k, sg[k], so[k]) for k in ('rays_closest', 'rays_any', 'nodes_closest') if sg[k] != so[k]]}", flush=True)
    gpu.close(); orc.close()
print(f"deep seeds {first}..{first + count - 1}: {count - len(bad)} bit-equal, mismatches {bad}; {time.time() - t0:.0f} s")

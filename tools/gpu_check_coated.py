#!/usr/bin/env python3
"""GPU bring-up check for the SURVEY 8f rows built so far (Coated* materials, MixMaterial, BilinearPatch): render parity against the oracle. Prints, never asserts."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
from shimmer_amd import abi, scenes, render
import oracle_py

lib = abi.load_library()
for name, sc, spp in (("cornell coated", scenes.cornell_box(lib, 64, 64, coated=True), 8),
                      ("ganesha coated", scenes.ganesha_proxy(lib, 64, 64, n=32, coated=True), 4),
                      ("cornell mix", scenes.cornell_box(lib, 64, 64, mix=True), 8),
                      ("cornell patches", scenes.cornell_box(lib, 64, 64, patches=True), 8),
                      ("cornell patches skewed", scenes.cornell_box(lib, 48, 48, patches=True, patch_skew=2e-3), 4)):
    gpu, orc = render.Renderer(lib, sc.desc, 0), oracle_py.Oracle(sc.desc)
    p = render.make_params(seed=3, spp=spp, max_depth=5)
    t = time.time(); fg, sg = gpu.render(p); tg = time.time() - t
    t = time.time(); fo, so = orc.render(p, n_threads=os.cpu_count() or 1); to = time.time() - t
    a, b = render.film_to_rgb(fg), render.film_to_rgb(fo)
    print(f"[{name}] bit_exact={np.array_equal(fg, fo)} Linf={np.max(np.abs(a - b)):.3e} npix_diff={int((np.abs(a - b).max(axis=2) > 0).sum())} "
          f"mean={a.mean():.4f}/{b.mean():.4f} rays gpu={sg['rays_closest']} cpu={so['rays_closest']} gpu {tg:.2f}s cpu {to:.2f}s", flush=True)
    gpu.close(); orc.close()

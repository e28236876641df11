import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "oracle"))
import oracle_py
from shimmer_amd import abi, render, scenes
lib = abi.load_library()
for seed in (0, 1, 3):
    sc = scenes.random_scene(lib, seed)
    for depth in (0, 1, 7):
        p = render.make_params(seed=100 + seed, spp=6, max_depth=depth, regularize=bool(seed % 4 == 3))
        gpu, orc = render.Renderer(lib, sc.desc, 0), oracle_py.Oracle(sc.desc)
        fg, sg = gpu.render(p)
        fo, so = orc.render(p, n_threads=os.cpu_count() or 1)
        bad = (fg["rgb_sum"] != fo["rgb_sum"]).any(axis=-1)
        print("seed", seed, "depth", depth, "bad pixels", int(bad.sum()), "of", bad.size, "gpu zero", int((fg["rgb_sum"] == 0).all(axis=-1).sum()),
              {k: (sg[k], so[k]) for k in ("rays_closest", "rays_any", "nodes_closest") if sg[k] != so[k]})
        if bad.any():
            ys, xs = np.nonzero(bad)
            print("   rows", ys.min(), ys.max(), "cols", xs.min(), xs.max())
        gpu.close(); orc.close()

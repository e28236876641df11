"""Ad-hoc larger parity sweep (more samples -> rarer branches): GPU film vs oracle film, bit for bit."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle_py
from shimmer_amd import abi, render, scenes
lib = abi.load_library()
cases = [("fuzz%d" % s, (lambda s=s: scenes.random_scene(lib, s, 128, 96)), 24, 8, {}) for s in (12, 13, 14, 15, 5, 6)]
cases += [("textured", lambda: scenes.cornell_box(lib, 256, 256, textured=True), 24, 6, {}),
          ("textured_nocoat_ewa", lambda: scenes.cornell_box(lib, 192, 192, textured=True, textured_coated_ceiling=False, texture_filter="ewa"), 16, 6, {}),
          ("instanced", lambda: scenes.instanced_scene(lib, 256, 192), 32, 6, {}),
          ("environment", lambda: scenes.three_spheres(lib, 256, 192, camera=(0.75, 0.5, 9.0), environment=scenes.environment_image(64)), 32, 5, {}),
          ("textured_forced", lambda: scenes.cornell_box(lib, 128, 128, textured=True), 16, 6, dict(force_diffuse=True)),
          ("instanced_simple", lambda: scenes.instanced_scene(lib, 128, 96), 16, 5, dict(integrator="simplepath")),
          ("fuzz13_randomwalk", lambda: scenes.random_scene(lib, 13, 96, 64), 32, 6, dict(integrator="randomwalk"))]
bad_total = 0
for name, mk, spp, depth, kw in cases:
    sc = mk()
    p = render.make_params(seed=21, spp=spp * int(os.environ.get("SPP_SCALE", "1")), max_depth=depth, **kw)
    gpu, orc = render.Renderer(lib, sc.desc, 0), oracle_py.Oracle(sc.desc)
    t0 = time.time(); fg, sg = gpu.render(p); t1 = time.time()
    fo, so = orc.render(p, n_threads=os.cpu_count() or 1); t2 = time.time()
    bad = int((fg["rgb_sum"] != fo["rgb_sum"]).any(axis=-1).sum())
    cnt = all(sg[k] == so[k] for k in ("rays_closest", "rays_any", "nodes_closest", "tris_closest", "nodes_any", "tris_any"))
    nan = int(np.isnan(fo["rgb_sum"]).sum())
    print(f"{name:22s} paths {sg['paths']:9d} rays {sg['rays_closest'] + sg['rays_any']:10d} differing pixels {bad} counters_equal {cnt} nan {nan} gpu {t1 - t0:.2f}s cpu {t2 - t1:.2f}s", flush=True)
    bad_total += bad + (0 if cnt else 1)
    gpu.close(); orc.close()
print("TOTAL MISMATCHES", bad_total)

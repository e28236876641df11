#!/usr/bin/env python3
"""Quick GPU bring-up check: trace + render parity against the oracle on small scenes. Prints, never asserts."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
from shimmer_amd import abi, scenes, render
import oracle_py

lib = abi.load_library()
print("devices:", lib.shm_device_count(), flush=True)

def rays_for(sc, n, seed=42):
    rng = np.random.default_rng(seed)
    b = sc.info["bounds"]
    lo, hi = b[:, :3].min(0), b[:, 3:].max(0)
    c, r = (lo + hi) / 2, np.linalg.norm(hi - lo) / 2
    o = c + (rng.random((n, 3)) * 2 - 1) * r * 1.2
    d = rng.normal(size=(n, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = np.zeros((n, 8), np.float32)
    rays[:, :3], rays[:, 3:6], rays[:, 6] = o, d, np.inf
    return rays

def check_scene(name, sc, spp, max_depth, nrays=20000):
    t0 = time.time()
    gpu = render.Renderer(lib, sc.desc, 0)
    orc = oracle_py.Oracle(sc.desc)
    rays = rays_for(sc, nrays)
    hg, sg = gpu.trace(rays)
    ho, so = orc.trace(rays)
    same = all(np.array_equal(hg[k], ho[k]) for k in ["prim", "t", "b0", "b1", "b2", "phi"])
    print(f"[{name}] closest: bitwise={same} hits={int((hg['prim']>=0).sum())}/{nrays} nodes gpu={sg['nodes_closest']} cpu={so['nodes_closest']} tris gpu={sg['tris_closest']} cpu={so['tris_closest']}", flush=True)
    if not same:
        bad = np.nonzero((hg["prim"] != ho["prim"]) | (hg["t"] != ho["t"]))[0][:5]
        for i in bad: print("   ", i, hg[i], ho[i])
    rays[:, 6] = 2.0
    ag, s2 = gpu.trace(rays, any_hit=True)
    ao, s3 = orc.trace(rays, any_hit=True)
    print(f"[{name}] any: equal={np.array_equal(ag, ao)} occluded={int(ag.sum())} nodes gpu={s2['nodes_any']} cpu={s3['nodes_any']}", flush=True)
    params = render.make_params(seed=3, spp=spp, max_depth=max_depth)
    t1 = time.time()
    fg, stg = gpu.render(params)
    t2 = time.time()
    fo, sto = orc.render(params, n_threads=os.cpu_count())
    t3 = time.time()
    a, b = render.film_to_rgb(fg), render.film_to_rgb(fo)
    d = np.abs(a - b)
    print(f"[{name}] render {sc.desc.film.full_resolution[0]}x{sc.desc.film.full_resolution[1]}x{spp}: bit_exact={np.array_equal(fg, fo)} Linf={d.max():.3e} "
          f"npix_diff={int((d.max(axis=2) > 0).sum())} mean={a.mean():.4f}/{b.mean():.4f} gpu {t2-t1:.2f}s cpu {t3-t2:.2f}s", flush=True)
    for k in ["paths", "rays_closest", "rays_any", "nodes_closest", "tris_closest", "nodes_any", "tris_any"]:
        if stg[k] != sto[k]: print(f"    counter {k}: gpu={stg[k]} cpu={sto[k]}")
    print(f"    gpu ms: total={stg['ms_total']:.2f} closest={stg['ms_trace_closest']:.2f} any={stg['ms_trace_any']:.2f} shade={stg['ms_shade']:.2f}", flush=True)
    gpu.close(); orc.close()

which = sys.argv[1:] or ["s2", "s1", "s3s", "s4s"]
if "s2" in which: check_scene("S2 cornell", scenes.cornell_box(lib, 96, 96), 16, 5)
if "s1" in which: check_scene("S1 sphere", scenes.sphere_light(lib, 64, 64), 8, 5)
if "s3s" in which: check_scene("S3 small", scenes.ganesha_proxy(lib, 96, 96, n=40), 8, 5)
if "s4s" in which: check_scene("S4 small", scenes.crown_proxy(lib, 60, 84, level=2, n_glass=12, n_gold=4), 8, 32)

#!/usr/bin/env python3
"""Sweep the traversal kernels' thresholds on the headline frame with ONE scene build (the knobs are read at shm_scene_create).
Usage: python tools/sweep_trace.py --refill 8 12 16 24 --leaf 8 12 16 24 --leaf-any 4 8 12 [--spp 256]"""
import argparse, itertools, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--refill", type=int, nargs="*", default=[16])
ap.add_argument("--leaf", type=int, nargs="*", default=[16])
ap.add_argument("--leaf-any", type=int, nargs="*", default=[8])
ap.add_argument("--refill-any", type=int, nargs="*", default=[0], help="0 = the same as --refill")
ap.add_argument("--spp", type=int, default=256)
ap.add_argument("--res", type=int, default=1024)
args = ap.parse_args()
from shimmer_amd import abi, scenes, render
lib = abi.load_library()
sc = scenes.ganesha_proxy(lib, args.res, args.res)
p = render.make_params(seed=0, spp=args.spp, max_depth=5)
for r, l, la, ra in itertools.product(args.refill, args.leaf, args.leaf_any, args.refill_any):
    os.environ.update(SHM_REFILL_MIN=str(r), SHM_LEAF_MIN=str(l), SHM_LEAF_MIN_ANY=str(la), SHM_REFILL_MIN_ANY=str(ra or r))
    rr = render.Renderer(lib, sc.desc, 0)
    rr.clear(); rr.render_device(p)
    best = None
    for _ in range(2):
        rr.clear()
        st = rr.render_device(p)
        if best is None or st["ms_total"] < best["ms_total"]:
            best = st
    rr.close()
    print(f"refill_min={r:2d} refill_min_any={ra or r:2d} leaf_min={l:2d} leaf_min_any={la:2d}: total {best['ms_total']:7.1f} ms  closest {best['ms_trace_closest']:6.1f}  any {best['ms_trace_any']:6.1f}  shade {best['ms_shade']:6.1f}", flush=True)

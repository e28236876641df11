#!/usr/bin/env python3
"""A/B of scene-level knobs on ONE box: every configuration ("K=V,K2=V2", "" = defaults) renders the same frames in this process — the knobs are
read when a scene is created —, the films must be BIT-equal across configurations, times are printed per configuration, alternating, `--rounds` times.

    python tools/film_ab.py --scenes S3,C4 --rounds 2 "" "SHM_ANY_ORDER_FREE=0" "SHM_LEAF_MIN_FAST=32"
    python tools/film_ab.py --scenes S3small,C4small,C2 --check-only "" "SHM_ANY_ORDER_FREE=0"      (small frames: film equality only)
"""
import argparse, os, sys, time, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from shimmer_amd import abi, scenes, render

ap = argparse.ArgumentParser()
ap.add_argument("--scenes", default="S3")
ap.add_argument("--rounds", type=int, default=2)
ap.add_argument("--check-only", action="store_true")
ap.add_argument("configs", nargs="*", default=[""])
args = ap.parse_args()
lib = abi.load_library()
SC = {
    "S3": (lambda: scenes.ganesha_proxy(lib, 1024, 1024), 256, 5),
    "S3c": (lambda: scenes.ganesha_proxy(lib, 1024, 1024, coated=True), 256, 5),
    "S3p": (lambda: scenes.ganesha_proxy(lib, 1024, 1024, variant="patch_emitter"), 256, 5),
    "S3s": (lambda: scenes.ganesha_proxy(lib, 1024, 1024, variant="one_sphere"), 256, 5),
    "S3i": (lambda: scenes.ganesha_proxy(lib, 1024, 1024, variant="instanced"), 256, 5),
    "S3t": (lambda: scenes.ganesha_proxy(lib, 1024, 1024, variant="textured_floor"), 256, 5),
    "S3ig": (lambda: scenes.ganesha_proxy(lib, 1024, 1024, variant="instance_grid"), 256, 5),
    "S3igsmall": (lambda: scenes.ganesha_proxy(lib, 256, 256, n=120, variant="instance_grid"), 16, 5),
    "S3q": (lambda: scenes.ganesha_proxy(lib, 1024, 1024, variant="quads"), 256, 5),
    "S3au": (lambda: scenes.ganesha_proxy(lib, 1024, 1024, object_material="gold"), 256, 5),
    "S3gl": (lambda: scenes.ganesha_proxy(lib, 1024, 1024, object_material="glass"), 256, 5),
    "S3gl16": (lambda: scenes.ganesha_proxy(lib, 1024, 1024, object_material="glass"), 256, 16),
    "S3cc": (lambda: scenes.ganesha_proxy(lib, 1024, 1024, object_material="coated_conductor"), 256, 5),
    "S3to": (lambda: scenes.ganesha_proxy(lib, 1024, 1024, variant="textured_object"), 256, 5),
    "S3m": (lambda: scenes.ganesha_proxy(lib, 1024, 1024, variant="mesh_emitter"), 256, 5),
    "S3n": (lambda: scenes.ganesha_proxy(lib, 1024, 1024, variant="smooth"), 256, 5),
    "S3nc": (lambda: scenes.ganesha_proxy(lib, 1024, 1024, variant="smooth", coated=True), 256, 5),
    "S3q50": (lambda: scenes.ganesha_proxy(lib, 1024, 1024, quad_fraction=0.5), 256, 5),
    "S3q25": (lambda: scenes.ganesha_proxy(lib, 1024, 1024, quad_fraction=0.25), 256, 5),
    "S3q10": (lambda: scenes.ganesha_proxy(lib, 1024, 1024, quad_fraction=0.10), 256, 5),
    "S3q03": (lambda: scenes.ganesha_proxy(lib, 1024, 1024, quad_fraction=0.03), 256, 5),
    "S3qc": (lambda: scenes.ganesha_proxy(lib, 1024, 1024, variant="quads", coated=True), 256, 5),
    "C4": (lambda: scenes.crown_proxy(lib, 1000, 1400), 256, 32),
    "C2": (lambda: scenes.cornell_box(lib, 512, 512), 64, 5),
    "C2t": (lambda: scenes.cornell_box(lib, 512, 512, textured=True), 64, 6),
    "S3small": (lambda: scenes.ganesha_proxy(lib, 256, 256, n=120), 16, 5),
    "C4small": (lambda: scenes.crown_proxy(lib, 250, 350, level=3, n_glass=24, n_gold=8), 16, 32),
}
KEYS = ("paths", "rays_closest", "rays_any", "nodes_closest", "tris_closest")
for name in args.scenes.split(","):
    make, spp, depth = SC[name]
    sc = make()
    ref = None
    p = render.make_params(seed=0, spp=spp, max_depth=depth)
    for rnd in range(1 if args.check_only else args.rounds):
        for cfg in args.configs:
            saved = {}
            for kv in filter(None, cfg.split(",")):
                k, v = kv.split("=")
                saved[k] = os.environ.get(k)
                os.environ[k] = v
            r = render.Renderer(lib, sc.desc, 0)
            for k, v in saved.items():
                if v is None:
                    os.environ.pop(k)
                else:
                    os.environ[k] = v
            r.clear(); r.render_device(p)
            r.clear()
            t0 = time.perf_counter()
            st = r.render_device(p)
            dt = time.perf_counter() - t0
            line = ""
            if rnd == 0:
                film = r.read_film()
                h = hashlib.sha256(film.tobytes()).hexdigest()[:16]
                cnt = tuple(st[k] for k in KEYS)
                if ref is None:
                    ref = (h, cnt)
                ok = (h, cnt) == ref
                line = f" film {h} {'== first' if ok else '!= FIRST  <<<<<<<< MISMATCH'} nodes_closest/ray {st['nodes_closest'] / max(1, st['rays_closest']):.2f} nodes_any/ray {st['nodes_any'] / max(1, st['rays_any']):.2f} tris_any/ray {st['tris_any'] / max(1, st['rays_any']):.2f}"
            rays = st["rays_closest"] + st["rays_any"]
            print(f"{name:8s} [{cfg or 'defaults':40s}] {rays / dt / 1e6:8.1f} Mray/s {dt * 1e3:8.1f} ms | closest {st['ms_trace_closest']:7.1f} any {st['ms_trace_any']:7.1f} shade {st['ms_shade']:7.1f}{line}", flush=True)
            r.close()

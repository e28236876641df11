#!/usr/bin/env python3
"""Experiment: does the headline frame gain from TWO half-frames in flight on one GPU?

The staged kernels of one frame alternate between a VALU-bound persistent traversal (8 waves per SIMD) and a latency-bound shade kernel
(2 waves per SIMD, 205 VGPRs). Two scenes (same description, own workspace and streams each) render the two halves of the tile list from two
host threads; the hardware scheduler is free to run one half's shade beside the other half's traversal. Compared with one scene rendering
every tile. Prints Mray/s for both and the ratio.
    python tools/exp_two_streams.py [--spp 256] [--res 1024] [--reps 2]
Env knobs of the library apply to both (SHM_TRACE3_BLOCKS_PER_CU=4 leaves half of every SIMD's wave slots to the other half-frame).
"""
import argparse
import sys
import threading
import time

import numpy as np
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--spp", type=int, default=256)
    ap.add_argument("--res", type=int, default=1024)
    ap.add_argument("--n", type=int, default=599)
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--coated", action="store_true")
    ap.add_argument("--parts", type=int, default=2)
    ap.add_argument("--sequential", action="store_true", help="run the parts one after the other (control)")
    args = ap.parse_args()
    from shimmer_amd import abi, scenes, render
    lib = abi.load_library()
    sc = scenes.ganesha_proxy(lib, args.res, args.res, n=args.n, coated=args.coated)
    params = render.make_params(seed=1, spp=args.spp, max_depth=5)
    one = render.Renderer(lib, sc.desc, device=0)

    def rays(st):
        return st["rays_closest"] + st["rays_any"]

    def sync():
        abi.check(lib, lib.shm_device_synchronize(0), "sync")

    # --- one scene, every tile ---
    one.clear(); one.render_device(params); sync()
    t0 = time.perf_counter()
    n_rays = 0
    for _ in range(args.reps):
        one.clear()
        n_rays += rays(one.render_device(params))
    sync()
    dt1 = time.perf_counter() - t0
    print(f"one scene, all tiles        : {n_rays / dt1 / 1e6:8.1f} Mray/s  ({dt1 / args.reps * 1e3:.1f} ms per frame)", flush=True)
    film_one = np.array(one.read_film()).view(np.float64).copy()

    # --- `parts` scenes, interleaved tile shards, one host thread each ---
    rs = [one] + [render.Renderer(lib, sc.desc, device=0) for _ in range(args.parts - 1)]
    shards = [render.shard_tiles(one.n_tiles, one.tiles_per_row, k, args.parts, lib=lib) for k in range(args.parts)]
    out = [0] * args.parts

    def work(k):
        tot = 0
        for _ in range(args.reps):
            rs[k].clear()
            tot += rays(rs[k].render_device(params, shards[k]))
        out[k] = tot

    for k in range(args.parts):  # warm-up (workspace allocation)
        rs[k].clear(); rs[k].render_device(params, shards[k])
    sync()
    t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(k,)) for k in range(args.parts)]
    if args.sequential:
        for t in th: t.start(); t.join()
    else:
        for t in th: t.start()
        for t in th: t.join()
    sync()
    dt2 = time.perf_counter() - t0
    print(f"{args.parts} scenes, tile shards, threads: {sum(out) / dt2 / 1e6:8.1f} Mray/s  ({dt2 / args.reps * 1e3:.1f} ms per frame)   x{dt1 / dt2:.3f}", flush=True)
    # the two half films add up to the whole one (disjoint pixels)
    films = [np.array(r.read_film()).view(np.float64) for r in rs]
    total = np.zeros_like(film_one)
    for f in films:
        total = total + f
    print("films identical:", bool(np.array_equal(total.view(np.uint8), film_one.view(np.uint8))) if args.parts == 1 else
          f"max |sum of parts - whole| = {np.abs(total - film_one).max()}")


if __name__ == "__main__":
    main()

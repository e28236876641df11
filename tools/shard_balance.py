#!/usr/bin/env python3
"""Load balance of the C5 tile sharding (BASELINE configs[4]: S3 at 3840x2160, 1024 spp, 8 ranks) measured on ONE GPU: every rank's tile set
(shm_shard_tiles) is rendered in turn and timed, so max / mean of the per-rank times predicts what an 8-GPU run loses to imbalance (the
film gather adds 33 MB per peer). Usage: python tools/shard_balance.py [--world 8] [--spp 1024] [--blocks B ...]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--world", type=int, default=8)
ap.add_argument("--spp", type=int, default=1024)
ap.add_argument("--width", type=int, default=3840)
ap.add_argument("--height", type=int, default=2160)
ap.add_argument("--blocks", type=int, nargs="*", default=[0], help="SHM_SHARD_BLOCKS values to compare (0 = the library's default)")
args = ap.parse_args()
from shimmer_amd import abi, scenes, render
lib = abi.load_library()
sc = scenes.ganesha_proxy(lib, args.width, args.height)
r = render.Renderer(lib, sc.desc, 0)
p = render.make_params(seed=0, spp=args.spp, max_depth=5)
r.clear(); r.render_device(p, render.shard_tiles(r.n_tiles, r.tiles_per_row, 0, args.world, lib=lib))  # warm-up: workspace allocation
for b in args.blocks:
    if b:
        os.environ["SHM_SHARD_BLOCKS"] = str(b)
    else:
        os.environ.pop("SHM_SHARD_BLOCKS", None)
    ms, rays = [], []
    for k in range(args.world):
        tiles = render.shard_tiles(r.n_tiles, r.tiles_per_row, k, args.world, lib=lib)
        r.clear()
        t0 = time.perf_counter()
        st = r.render_device(p, tiles)
        ms.append((time.perf_counter() - t0) * 1e3)
        rays.append(st["rays_closest"] + st["rays_any"])
    mean = sum(ms) / len(ms)
    print(f"blocks/rank {b or 'default'}: per-rank ms " + " ".join(f"{m:7.1f}" for m in ms) + f" | mean {mean:7.1f} max {max(ms):7.1f} -> balance efficiency {mean / max(ms):.3f}; "
          f"whole frame {sum(rays) / 1e9:.2f} G rays, ideal {sum(rays) / (max(ms) * 1e-3) / 1e6:.0f} Mray/s on {args.world} GPUs", flush=True)
r.close()

#!/usr/bin/env python3
"""bench.py — Mray/s (primary + secondary rays) of the wavefront path tracer on the BASELINE.json headline config:
S3 "ganesha-proxy" (4 305 626 primitives, 8.52 M BVH nodes), 1024x1024, 256 spp, maxdepth 5, on N MI355X of one node.

A step = one whole ImageTileIntegrator::render of that frame (all spp-waves 1,1,2,...,64,64,64 over all 8x8 tiles).
Scene arrays are resident in HBM before the timed region; tiles are sharded across ranks (no collective while
rendering) and the film slabs are gathered to rank 0 over RCCL inside the timed region.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--spp S] [--res R] [--no-cpu-baseline]
N > 1 is launched by the driver through torch.distributed.run (RANK / LOCAL_RANK / WORLD_SIZE in the env).
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured-achievable)


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def pmc_traffic(args):
    """HBM traffic of the K2 kernel per launch from the committed rocprofv3 PMC passes of THIS configuration
    (tools/profile_gpu.sh -> profiles/*.traffic.json): 2 x FETCH_SIZE (the gfx950 correction of MI355X_MICROARCH.md
    section HBM) + WRITE_SIZE, in bytes. None when no profile of this configuration is committed."""
    f = ROOT / "profiles" / f"traffic_res{args.res}_spp{args.spp}_n{args.n}_depth{args.max_depth}.json"
    if not f.exists():
        return None
    t = json.loads(f.read_text()).get("k_trace3<closest>")
    if not t:
        return None
    return 2.0 * t.get("FETCH_SIZE_bytes_per_dispatch_raw", 0.0) + t.get("WRITE_SIZE_bytes_per_dispatch_raw", 0.0)


def cpu_baseline(sc, params_full, host_lib, budget_s=15.0):
    """The CPU oracle (kind "port": a C++ restatement of the reference loop, oracle/oracle.cpp) timed on this box's host
    cores on a bounded sample of the same workload: a centred crop of the frame at a reduced spp."""
    sys.path.insert(0, str(ROOT / "oracle"))
    import oracle_py
    from shimmer_amd import render, scene as scn

    cores = os.cpu_count() or 1
    orc = oracle_py.Oracle(sc.desc)
    pb = orc.pixel_bounds
    w, h = pb[2] - pb[0], pb[3] - pb[1]

    def crop_tiles(side):
        cw, ch = min(w, side), min(h, side)
        x0, y0 = pb[0] + (w - cw) // 2, pb[1] + (h - ch) // 2
        return (cw, ch) + scn.tiles_for(host_lib, (x0, y0, x0 + cw, y0 + ch))

    def run(side, spp):
        cw, ch, tiles, n_tiles = crop_tiles(side)
        p = render.make_params(seed=params_full.seed, spp=spp, max_depth=params_full.max_depth)
        t0 = time.perf_counter()
        _, st = orc.render(p, n_threads=cores, tiles=tiles, n_tiles=n_tiles)
        return cw, ch, st["rays_closest"] + st["rays_any"], time.perf_counter() - t0

    # probe (1 and 5 spp on the 256^2 crop) to size the sample to ~budget_s seconds of wall time on all host cores
    run(256, 1)  # thread start-up, page faults
    _, _, _, t0 = run(256, 1)
    _, _, rays1, t1 = run(256, 5)
    full = params_full.samples_per_pixel
    want = budget_s / max((t1 - t0) / 4.0, 1e-4)  # spp the 256^2 crop could take (fixed per-call cost cancels)
    side, spp = 256, int(max(1, min(full, want)))
    if want > full:
        side, spp = 512, int(max(1, min(full, want / 4.0)))
    cw, ch, rays, dt = run(side, spp)
    orc.close()
    model = ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"value": rays / dt / 1e6, "unit": "Mray/s", "cores": cores, "kind": "port",
            "sample": f"oracle/oracle.cpp, {cores} threads over 8x8 tiles, centred {cw}x{ch} crop of the same frame at {spp} spp "
                      f"({rays} rays in {dt:.1f} s; probe {rays1} rays in {t1:.2f} s); cpu: {model}"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--spp", type=int, default=256)
    ap.add_argument("--res", type=int, default=1024)
    ap.add_argument("--width", type=int, default=0, help="non-square frames (e.g. the 3840x2160 of BASELINE config C5): overrides --res")
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--n", type=int, default=599, help="cube-sphere subdivision (599 -> 4 305 612 triangles)")
    ap.add_argument("--max-depth", type=int, default=5)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--coated", action="store_true", help="S3 with a CoatedDiffuse object (LayeredBxDF, SURVEY 8f-1) instead of the "
                    "headline diffuse one: a side measurement, not the BASELINE config")
    ap.add_argument("--shard-of", type=int, default=0, help="development: render only the tiles rank 0 of N would own (no gather), "
                    "to estimate the per-rank time of an N-GPU run on one GPU")
    ap.add_argument("--shard-rank", type=int, default=0, help="development: which rank's tiles --shard-of renders")
    ap.add_argument("--force-dist", action="store_true", help="exercise the RCCL init + film gather path even with one rank")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run (one process per GPU)")
        args.gpus = world

    import numpy as np
    import torch
    import torch.distributed as dist
    from shimmer_amd import abi, scenes, render

    lib = abi.load_library()
    if lib.shm_device_count() < 1 or not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the render path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    t0 = time.perf_counter()
    width, height = (args.width or args.res), (args.height or args.res)
    sc = scenes.ganesha_proxy(lib, width, height, n=args.n, coated=args.coated)
    t_scene = time.perf_counter() - t0
    t0 = time.perf_counter()
    r = render.Renderer(lib, sc.desc, device=local_rank)
    t_upload = time.perf_counter() - t0
    if rank == 0:
        log(f"[bench] scene {sc.name}: {sc.info['n_primitives']} prims, {sc.info['n_nodes']} nodes; build {t_scene:.1f}s, upload {t_upload:.2f}s")
    params = render.make_params(seed=args.seed, spp=args.spp, max_depth=args.max_depth)
    my_tiles = None if world == 1 else render.shard_tiles(r.n_tiles, r.tiles_per_row, rank, world)
    if world == 1 and args.shard_of > 1:
        my_tiles = render.shard_tiles(r.n_tiles, r.tiles_per_row, args.shard_rank, args.shard_of)

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def step():
        r.clear()
        st = r.render_device(params, my_tiles)
        film = None
        if use_dist:
            film = render.gather_film(render.film_tensor(r, device), rank, world, r.height, r.width, to_host=False)
            torch.cuda.synchronize(device)  # the send must have read this rank's film before the next step clears it
        return st, film

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    acc = {}
    for _ in range(args.steps):
        st, film = step()
        for k, v in st.items():
            acc[k] = acc.get(k, 0) + v
    barrier()
    dt = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        keys = ["paths", "rays_closest", "rays_any", "nodes_closest", "tris_closest", "nodes_any", "tris_any"]
        c = torch.tensor([acc[k] for k in keys], dtype=torch.float64, device=device)
        dist.all_reduce(c, op=dist.ReduceOp.SUM)
        tot = {k: float(v) for k, v in zip(keys, c.tolist())}
    else:
        tot = {k: float(v) for k, v in acc.items()}

    if rank == 0:
        rays = tot["rays_closest"] + tot["rays_any"]
        value = rays / dt / 1e6
        # roofline of the dominant kernel (K2 trace_closest) on rank 0, from HIP events on the library's render stream:
        # algorithmic bytes = 32 B per node visited + 48 B per primitive tested + (32 B ray read + 16 B hit write) per ray
        # (SURVEY §8d), summed over this rank's launches, over the summed launch durations (= per-launch averages' ratio).
        bytes_alg = 32.0 * acc["nodes_closest"] + 48.0 * acc["tris_closest"] + 48.0 * acc["rays_closest"]
        ms = acc["ms_trace_closest"]
        achieved = bytes_alg / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
        launches = max(1, acc["launches_closest"])
        out = {
            "metric": "Mray/s (primary+secondary) at 1024^2 256spp Ganesha; 1/2/4/8-GPU scaling",
            "value": value, "unit": "Mray/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{'S3c coated ' if args.coated else 'S3 '}ganesha-proxy ({sc.info['n_primitives']} prims, {sc.info['n_nodes']} BVH nodes), "
                                   f"{width}x{height}, {args.spp} spp, maxdepth {args.max_depth}, path integrator",
                       "tiles": "8x8, sharded across ranks in interleaved blocks of tile rows (~8 blocks per rank)" if world > 1 else "8x8",
                       "rays_per_step": rays / args.steps, "paths_per_step": tot["paths"] / args.steps,
                       "film_gather": "RCCL gather to rank 0 (inside the timed region)" if world > 1 else "none"},
            "roofline": {"bound": "hbm", "kernel": "k_trace3<closest> (BvhAggregate::intersect)", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": pmc_traffic(args) if world == 1 else None,
                         "bytes_per_launch": bytes_alg / launches, "avg_launch_ms": ms / launches, "launches": launches,
                         "nodes_per_ray": acc["nodes_closest"] / max(1, acc["rays_closest"]),
                         "prims_per_ray": acc["tris_closest"] / max(1, acc["rays_closest"]),
                         "closest_Mray_s_in_kernel": acc["rays_closest"] / (ms * 1e-3) / 1e6 if ms > 0 else 0.0},
            # the second traversal kernel, same accounting (informational; `roofline` above is the dominant kernel)
            "roofline_any_hit": {"kernel": "k_trace3<any> (BvhAggregate::intersect_predicate)",
                                 "achieved": ((32.0 * acc["nodes_any"] + 48.0 * acc["tris_any"] + 48.0 * acc["rays_any"]) / (acc["ms_trace_any"] * 1e-3) / 1e9)
                                 if acc["ms_trace_any"] > 0 else 0.0, "unit": "GB/s",
                                 "nodes_per_ray": acc["nodes_any"] / max(1, acc["rays_any"]), "prims_per_ray": acc["tris_any"] / max(1, acc["rays_any"]),
                                 "any_Mray_s_in_kernel": acc["rays_any"] / (acc["ms_trace_any"] * 1e-3) / 1e6 if acc["ms_trace_any"] > 0 else 0.0},
            "breakdown_ms_per_step": {"trace_closest": acc["ms_trace_closest"] / args.steps, "trace_any": acc["ms_trace_any"] / args.steps,
                                      "shade_generate_film": acc["ms_shade"] / args.steps, "gpu_total": acc["ms_total"] / args.steps},
        }
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(sc, params, lib)
            except Exception as e:  # the baseline is reporting only; never let it hide the GPU number
                out["cpu_baseline"] = {"value": None, "unit": "Mray/s", "cores": os.cpu_count(), "kind": "port", "sample": f"failed: {e}"}
        result_line = json.dumps(out)
    if use_dist and rank == 0 and film is not None:
        # the gathered film must equal what this rank's library holds when it is the only rank (self-check of the gather)
        host = film.cpu().numpy().view(render.FILM_DTYPE).reshape(r.height, r.width)
        if world == 1 and not np.array_equal(host, r.read_film()):
            raise SystemExit("film gather mismatch")
        if not (host["weight_sum"] == float(args.spp)).all():  # every pixel of the frame received all its samples, from some rank
            raise SystemExit("gathered film is incomplete")
    if rank == 0 and world == 1 and not args.shard_of:
        # self-check outside the timed region: every pixel received all its samples, all sums finite
        host = r.read_film()
        if not (host["weight_sum"] == float(args.spp)).all() or not np.isfinite(host["rgb_sum"]).all():
            raise SystemExit("film is incomplete or not finite")
    r.close()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        # RCCL prints its version banner through C stdio, which is flushed at exit: flush it now so that the JSON line is the
        # last thing on stdout
        import ctypes
        ctypes.CDLL(None).fflush(None)
        print(result_line, flush=True)


if __name__ == "__main__":
    main()

"""The N>1 path on CPU: two gloo processes shard the tile list exactly as bench.py does (render.shard_tiles), render
their shards (the CPU oracle stands in for the GPU renderer, which does not exist here) and gather the film slabs with
render.gather_film; the result must equal the single-process film bit for bit."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parents[1]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, str(ROOT))
    sys.path.insert(0, str(ROOT / "oracle"))
    import torch
    import torch.distributed as dist
    import oracle_py
    from shimmer_amd import abi, render, scene as scn, scenes

    dist.init_process_group("gloo", rank=rank, world_size=world)
    lib = abi.load_library()
    sc = scenes.cornell_box(lib, 64, 40)
    o = oracle_py.Oracle(sc.desc)
    tiles, n = scn.tiles_for(lib, o.pixel_bounds)
    mine = render.shard_tiles(n, (o.width + 7) // 8, rank, world, rows_per_block=2)
    sub = (abi.ShmTile * max(1, len(mine)))(*[tiles[int(i)] for i in mine])
    p = render.make_params(seed=9, spp=4, max_depth=5)
    film, _ = o.render(p, n_threads=2, tiles=sub, n_tiles=len(mine))
    local = torch.from_numpy(film.view(np.float64).reshape(-1).copy())
    total = render.gather_film(local, rank, world, o.height, o.width)
    if rank == 0:
        whole, _ = o.render(p, n_threads=2)
        np.save(out_path, np.array([int(np.array_equal(total, whole)), int(len(mine)), int(n)]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_tile_sharding_and_film_gather(tmp_path):
    import torch.multiprocessing as mp
    port = _free_port()
    out = tmp_path / "r.npy"
    mp.spawn(_worker, args=(2, port, str(out)), nprocs=2, join=True)
    ok, n_mine, n_all = np.load(out)
    assert ok == 1 and 0 < n_mine < n_all


def test_shard_tiles_partition():
    from shimmer_amd import render
    n_tiles, per_row = 128 * 128, 128
    parts = [render.shard_tiles(n_tiles, per_row, r, 8) for r in range(8)]
    allidx = np.sort(np.concatenate(parts))
    assert np.array_equal(allidx, np.arange(n_tiles))  # disjoint and complete
    assert max(len(p) for p in parts) == min(len(p) for p in parts)  # 128 tile rows in 1-row blocks, 16 blocks per rank
    assert len(np.unique(np.diff(parts[0]) > 1)) == 2  # interleaved, not one contiguous band

"""The N>1 path with the REAL renderer (run with `pytest -m gpu`): two processes on the one GPU of the test box shard the
tiles exactly as bench.py does (render.shard_tiles), render their shards with libshimmer_hip.so (shm_render_device),
and gather the film slabs with render.gather_film — over gloo, because RCCL refuses two ranks on one device; the RCCL
transport itself is exercised by `bench.py --force-dist`. The gathered film must equal the single-process GPU film AND the
oracle's, bit for bit."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
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
    from shimmer_amd import abi, render, scenes

    dist.init_process_group("gloo", rank=rank, world_size=world)
    lib = abi.load_library()
    sc = scenes.ganesha_proxy(lib, 96, 72, n=24)
    r = render.Renderer(lib, sc.desc, device=0)
    mine = render.shard_tiles(r.n_tiles, r.tiles_per_row, rank, world)
    p = render.make_params(seed=11, spp=8, max_depth=5)
    r.clear()
    st = r.render_device(p, mine)
    local = torch.from_numpy(r.read_film().view(np.float64).reshape(-1).copy())
    total = render.gather_film(local, rank, world, r.height, r.width)
    rays = torch.tensor([float(st["rays_closest"] + st["rays_any"])], dtype=torch.float64)
    dist.all_reduce(rays)
    if rank == 0:
        import oracle_py
        whole, sw = r.render(p)
        orc, _ = oracle_py.Oracle(sc.desc).render(p, n_threads=4)
        np.save(out_path, np.array([int(np.array_equal(total, whole)), int(np.array_equal(total, orc)), int(len(mine)), int(r.n_tiles),
                                    int(rays.item() == sw["rays_closest"] + sw["rays_any"])]))
    r.close()
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharded_gpu_render_equals_single(gpu_lib, tmp_path):
    import torch.multiprocessing as mp
    port = _free_port()
    out = tmp_path / "r.npy"
    mp.spawn(_worker, args=(2, port, str(out)), nprocs=2, join=True)
    same_as_single, same_as_oracle, n_mine, n_all, rays_ok = np.load(out)
    assert same_as_single == 1 and same_as_oracle == 1 and rays_ok == 1 and 0 < n_mine < n_all

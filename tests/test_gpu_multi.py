"""Multi-GPU behind the C ABI on the one-GPU test box (run with `pytest -m gpu`):
  * shm_render_multi with 1, 2 and 3 replicas (device ordinal 0 repeated: the replicas share the GPU, which exercises the host
    threads, the C++ tile sharding and the per-block peer copies of the film rows) == the single-scene film == the oracle's, bit for bit;
  * the one-process-per-GPU path: shm_dist_unique_id -> shm_dist_init (RCCL communicator, world 1) -> shm_render_sharded, and
    shm_dist_selftest, which pushes the film rows through the same ncclSend / ncclRecv group to the rank itself;
  * config C5's SHAPE (BASELINE.json configs[4]): the tiles rank r of 8 owns of a 3840x2160 frame, rendered by the GPU and by the
    oracle on the same tile list; and the torch harness's gather over the `nccl` backend with world_size 1.
The 8-rank run itself only happens at the driver's round end (RCCL refuses two ranks on one device)."""
import ctypes as C
import os

import numpy as np
import pytest

import oracle_py
from shimmer_amd import abi, render, scenes

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def small_s3(gpu_lib):
    sc = scenes.ganesha_proxy(gpu_lib, 160, 104, n=24)
    p = render.make_params(seed=5, spp=6, max_depth=5)
    r = render.Renderer(gpu_lib, sc.desc, device=0)
    film, stats = r.render(p)
    r.close()
    return sc, p, film, stats


@pytest.mark.parametrize("devices", [[0], [0, 0], [0, 0, 0]])
def test_render_multi_equals_single_device(gpu_lib, small_s3, devices):
    sc, p, film, stats = small_s3
    got, st = render.render_multi(gpu_lib, sc.desc, devices, p)
    assert np.array_equal(got, film)
    assert sum(s["rays_closest"] + s["rays_any"] for s in st) == stats["rays_closest"] + stats["rays_any"]
    assert sum(s["paths"] for s in st) == 160 * 104 * 6
    if len(devices) > 1:
        assert all(s["paths"] > 0 for s in st)  # every replica rendered a share
        moved = sum(s["gather_bytes"] for s in st)
        own0 = render.shard_tiles(13 * 20, 20, 0, len(devices))
        assert moved == (160 * 104 - sum(64 for _ in own0)) * 32  # every row the root does not own travelled exactly once


def test_render_multi_equals_oracle(gpu_lib, small_s3):
    sc, p, film, _ = small_s3
    o = oracle_py.Oracle(sc.desc)
    ref, _ = o.render(p, n_threads=os.cpu_count() or 4)
    o.close()
    got, _ = render.render_multi(gpu_lib, sc.desc, [0, 0], p)
    assert np.array_equal(got, ref)


def test_render_multi_rejects_bad_arguments(gpu_lib, small_s3):
    sc, p, _, _ = small_s3
    with pytest.raises(abi.ShimmerHipError):
        render.render_multi(gpu_lib, sc.desc, [99], p)  # no such device: reported, nothing unwinds
    with pytest.raises(abi.ShimmerHipError):
        render.render_multi(gpu_lib, sc.desc, [], p)


def test_dist_world1_rccl_render_and_loopback(gpu_lib, small_s3):
    """RCCL inside the library: communicator of one rank, shm_render_sharded (== the plain render), and the send / recv group
    looped back to the rank itself."""
    sc, p, film, stats = small_s3
    r = render.Renderer(gpu_lib, sc.desc, device=0)
    uid = r.dist_unique_id()
    assert len(uid) == abi.SHM_DIST_ID_BYTES and any(uid)
    r.dist_init(0, 1, uid)
    st = r.render_sharded(p)
    assert np.array_equal(r.read_film(), film)
    assert st["rays_closest"] == stats["rays_closest"] and st["gather_bytes"] == 0
    r.dist_selftest()
    st2 = r.render_sharded(p)  # the communicator survives a second frame
    assert st2["paths"] == st["paths"] and np.array_equal(r.read_film(), film)
    abi.check(gpu_lib, gpu_lib.shm_dist_finalize(r.handle), "shm_dist_finalize")
    r.close()


def test_dist_small_collectives_world1(gpu_lib, small_s3):
    """The collectives a host needs around shm_render_sharded (barrier, reductions, gather of per-rank figures) are the library's own:
    with a communicator of one rank they go through ncclAllReduce / ncclAllGather and must be the identity; without one they are too."""
    sc, p, _, _ = small_s3
    r = render.Renderer(gpu_lib, sc.desc, device=0)
    for with_comm in (False, True):
        if with_comm:
            r.dist_init(0, 1, r.dist_unique_id())
        r.dist_barrier()
        assert r.dist_allreduce([1.5, -2.0, 3e300], abi.SHM_REDUCE_MAX) == [1.5, -2.0, 3e300]
        assert r.dist_allreduce([0.1, 7.0], abi.SHM_REDUCE_SUM) == [0.1, 7.0]
        assert r.dist_allreduce([4.0], abi.SHM_REDUCE_MIN) == [4.0]
        assert r.dist_allgather([1.0, 2.0, 3.0]) == [[1.0, 2.0, 3.0]]
        info = r.dist_info()
        assert info["world"] == 1 and info["rank"] == 0
        assert info["rccl_ranks"] == (1 if with_comm else 0)
        assert info["rccl_version"] >= 20000 and "librccl" in info["librccl_path"] and "libamdhip64" in info["libamdhip_path"]
        if with_comm:
            assert info["rccl_device"] == 0 and info["n_my_tiles"] == r.n_tiles and info["rows_per_block"] >= 1
    with pytest.raises(abi.ShimmerHipError):
        r.dist_allreduce([0.0] * 5000)  # more than the control scratch holds: reported
    r.close()
    abi.check(gpu_lib, gpu_lib.shm_device_synchronize(0), "shm_device_synchronize")
    assert gpu_lib.shm_device_synchronize(99) != 0


def test_bench_self_launcher_world1_one_runtime_stack(gpu_lib):
    """`python bench.py --gpus 1 --launch --force-dist`: the parent starts ONE fresh rank process (no torch.distributed.run), which runs the
    whole N > 1 code path — id through the file store, shm_dist_init, barrier / max clock / counters through the library's RCCL
    collectives, shm_render_sharded, loop-back self test — on the HIP runtime and the RCCL the library is linked against, without torch."""
    import json
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parents[1]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "SHM_STORE_DIR")}
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "1", "--launch", "--force-dist", "--res", "128", "--spp", "8", "--n", "24", "--steps", "2",
                        "--warmup", "1", "--no-cpu-baseline", "--no-side"], cwd=root, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    out = json.loads([l for l in r.stdout.decode().splitlines() if l.startswith("{")][-1])
    rt = out["runtime"]
    assert rt["launcher"] == "shimmer_amd.launch" and rt["torch_imported"] is False
    assert rt["rccl_ranks"] == 1 and rt["librccl"].startswith("/opt/rocm") and rt["libamdhip64"].startswith("/opt/rocm")
    assert out["n_gpus"] == 1 and out["value"] > 0 and out["nonfinite_pixels"] == 0
    assert len(out["per_rank"]) == 1 and out["per_rank"][0]["tiles"] == 16 * 16 and out["per_rank"][0]["gather_MB"] == 0.0
    assert "self-launch: 1 rank process" in r.stderr.decode() and "librccl /opt/rocm" in r.stderr.decode()


def test_render_sharded_without_communicator_is_the_whole_frame(gpu_lib, small_s3):
    sc, p, film, _ = small_s3
    r = render.Renderer(gpu_lib, sc.desc, device=0)
    r.render_sharded(p)
    assert np.array_equal(r.read_film(), film)
    r.close()


@pytest.mark.parametrize("rank", [0, 3, 7])
def test_c5_shaped_shard_matches_oracle(gpu_lib, rank):
    """BASELINE configs[4] in shape: a 3840x2160 frame (129 600 tiles of 8x8, 480 per row), the shard rank r of 8 owns, 1 spp — the
    GPU film of those tiles equals the oracle's on the same tile list, and nothing outside them is touched."""
    sc = scenes.ganesha_proxy(gpu_lib, 3840, 2160, n=32)
    r = render.Renderer(gpu_lib, sc.desc, device=0)
    assert r.n_tiles == 129600 and r.tiles_per_row == 480
    mine = render.shard_tiles(r.n_tiles, r.tiles_per_row, rank, 8, lib=gpu_lib)
    assert abs(len(mine) - 129600 // 8) <= 480 * 4
    p = render.make_params(seed=3, spp=1, max_depth=5)
    r.clear()
    st = r.render_device(p, mine)
    film = r.read_film()
    r.close()
    assert st["paths"] == len(mine) * 64
    o = oracle_py.Oracle(sc.desc)
    sub = (abi.ShmTile * len(mine))(*[r.tiles[int(i)] for i in mine])
    ref, so = o.render(p, n_threads=os.cpu_count() or 4, tiles=sub, n_tiles=len(mine))
    o.close()
    assert np.array_equal(film, ref)
    assert st["rays_closest"] == so["rays_closest"] and st["nodes_closest"] == so["nodes_closest"] and st["tris_any"] == so["tris_any"]
    owned = np.zeros((2160, 3840), bool)
    for i in mine:
        t = r.tiles[int(i)]
        owned[t.y0:t.y1, t.x0:t.x1] = True
    assert (film["weight_sum"][owned] == 1.0).all() and (film["weight_sum"][~owned] == 0.0).all()


def test_c5_frame_at_full_size(gpu_lib):
    """BASELINE.json configs[4] ITSELF on one GPU: S3 (4 305 626 primitives) at 3840 x 2160, 1024 spp, maxdepth 5 — 8.49 G paths — through the
    path an 8-GPU run takes per rank: shm_dist_init (RCCL communicator, here of one rank) + shm_render_sharded. Size-independent properties
    (every pixel 1024 samples, finite, nothing gathered with one rank, run-to-run identity of film and counters, the send / recv loop-back),
    a 16 x 16 block on the object at all 1024 samples against the oracle — film and visit counters bit for bit —, and rank 3 of 8's REAL
    tile set on the real scene at 64 spp: owned pixels complete, the others untouched, a 1-in-97 sample of its tiles equal to the oracle's."""
    from shimmer_amd import scene as scn
    W, H, SPP = 3840, 2160, 1024
    sc = scenes.ganesha_proxy(gpu_lib, W, H)
    assert sc.info["n_primitives"] == 4305626
    p = render.make_params(seed=0, spp=SPP, max_depth=5)
    r = render.Renderer(gpu_lib, sc.desc, device=0)
    assert r.n_tiles == 129600 and r.tiles_per_row == 480
    r.dist_init(0, 1, r.dist_unique_id())
    s1 = r.render_sharded(p)
    f1 = r.read_film()
    s2 = r.render_sharded(p)
    assert np.array_equal(r.read_film(), f1)
    keys = ("paths", "rays_closest", "rays_any", "nodes_closest", "tris_closest", "nodes_any", "tris_any")
    for k in keys:
        assert s1[k] == s2[k], k
    assert s1["paths"] == W * H * SPP and s1["gather_bytes"] == 0
    assert (f1["weight_sum"] == float(SPP)).all() and np.isfinite(f1["rgb_sum"]).all() and (f1["rgb_sum"] >= 0).all()
    assert s1["rays_closest"] >= s1["paths"] and 0 < s1["rays_any"] <= s1["rays_closest"] and s1["rays_closest"] + s1["rays_any"] > 30_000_000_000
    r.dist_selftest()  # the film rows through the ncclSend / ncclRecv group, looped back, every byte compared
    # a 16 x 16 block in the object's silhouette (the 1024^2 headline test's block, scaled to this frame)
    x0, y0 = 1912, 928
    tiles, n = scn.tiles_for(gpu_lib, (x0, y0, x0 + 16, y0 + 16))
    orc = oracle_py.Oracle(sc.desc)
    fo, so = orc.render(p, n_threads=os.cpu_count() or 1, tiles=tiles, n_tiles=n)
    assert np.array_equal(f1[y0:y0 + 16, x0:x0 + 16], fo[y0:y0 + 16, x0:x0 + 16])
    assert so["nodes_closest"] / so["rays_closest"] > 20  # the block is on the object, not on a wall
    sel = np.array([i for i in range(r.n_tiles) if (lambda t: t.x0 >= x0 and t.x1 <= x0 + 16 and t.y0 >= y0 and t.y1 <= y0 + 16)(r.tiles[i])])
    assert len(sel) == n == 4
    r.clear()
    sb = r.render_device(p, tile_indices=sel)
    for k in keys:
        assert sb[k] == so[k], k
    # rank 3 of 8: its real tile set of this frame, 64 spp
    mine = render.shard_tiles(r.n_tiles, r.tiles_per_row, 3, 8, lib=gpu_lib)
    p64 = render.make_params(seed=0, spp=64, max_depth=5)
    r.clear()
    s3 = r.render_device(p64, mine)
    f3 = r.read_film()
    assert s3["paths"] == len(mine) * 64 * 64
    owned = np.zeros((H, W), bool)
    for i in mine:
        t = r.tiles[int(i)]
        owned[t.y0:t.y1, t.x0:t.x1] = True
    assert (f3["weight_sum"][owned] == 64.0).all() and (f3["weight_sum"][~owned] == 0.0).all() and np.isfinite(f3["rgb_sum"]).all()
    sample = mine[::97]
    sub = (abi.ShmTile * len(sample))(*[r.tiles[int(i)] for i in sample])
    fs, _ = orc.render(p64, n_threads=os.cpu_count() or 1, tiles=sub, n_tiles=len(sample))
    for i in sample:
        t = r.tiles[int(i)]
        assert np.array_equal(f3[t.y0:t.y1, t.x0:t.x1], fs[t.y0:t.y1, t.x0:t.x1]), int(i)
    orc.close()
    r.close()


def _nccl_world1_worker(rank, port, out_path):
    # a fresh process that imports torch FIRST (torch ships its own HIP runtime: it must be the one the process initialises)
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parents[1]
    sys.path.insert(0, str(root))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
    import torch
    import torch.distributed as dist
    from shimmer_amd import abi as abi_, render as render_, scenes as scenes_
    device = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=device)
    lib = abi_.load_library()
    sc = scenes_.ganesha_proxy(lib, 160, 104, n=24)
    p = render_.make_params(seed=5, spp=6, max_depth=5)
    r = render_.Renderer(lib, sc.desc, device=0)
    r.clear()
    r.render_device(p)
    total = render_.gather_film(render_.film_tensor(r, device), 0, 1, r.height, r.width)
    np.save(out_path, total.view(np.float64))
    r.close()
    dist.destroy_process_group()


def test_torch_harness_gather_over_nccl_world1(gpu_lib, small_s3, tmp_path):
    """The torch.distributed harness (render.gather_film) over backend `nccl` (= RCCL) with one rank: zero-copy view of the HBM film."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = tmp_path / "film.npy"
    mp.spawn(_nccl_world1_worker, args=(port, str(out)), nprocs=1, join=True)
    _, _, film, _ = small_s3
    assert np.array_equal(np.load(out), film.view(np.float64))

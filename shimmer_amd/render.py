"""Python mirror of the reference's render driver for the GPU path (src/render.rs:8-55 ->
ImageTileIntegrator::render, src/integrator.rs:226-322): spp-wave schedule over 8x8 tiles, film read-back,
RgbFilm::get_pixel_rgb (film.rs:720-738).  Multi-GPU: tiles are sharded across ranks (one process per GPU,
no data-path collective while rendering) and the film rows are gathered by RCCL INSIDE libshimmer_hip.so (shm_render_sharded);
the small collectives a host needs around it (barrier, reductions) are the library's too (Renderer.dist_*): no torch.
"""
import ctypes as C
import os

import numpy as np

from . import abi
from .scene import tiles_for, wave_schedule

FILM_DTYPE = np.dtype([("rgb_sum", "<f8", (3,)), ("weight_sum", "<f8")])


def make_params(seed=0, spp=4, max_depth=5, regularize=False, disable_pixel_jitter=False, disable_wavelength_jitter=False,
                integrator="path", sample_lights=True, sample_bsdf=True, force_diffuse=False, disable_texture_filtering=False, reference_quirks=True):
    p = abi.ShmRenderParams()
    p.integrator = {"path": abi.SHM_INTEGRATOR_PATH, "simplepath": abi.SHM_INTEGRATOR_SIMPLE_PATH,
                    "randomwalk": abi.SHM_INTEGRATOR_RANDOM_WALK}[integrator]
    p.sample_lights, p.sample_bsdf = int(sample_lights), int(sample_bsdf)
    p.force_diffuse, p.disable_texture_filtering = int(force_diffuse), int(disable_texture_filtering)
    p.disable_reference_quirks = 0 if reference_quirks else 1  # SHM_REFERENCE_QUIRKS: ON = reference-exact (the default)
    p.seed, p.samples_per_pixel, p.max_depth = seed, spp, max_depth
    p.regularize, p.disable_pixel_jitter, p.disable_wavelength_jitter = int(regularize), int(disable_pixel_jitter), int(disable_wavelength_jitter)
    return p


# XYZ -> linear sRGB (IEC 61966-2-1), the output_rgb_from_sensor_rgb of an sRGB film behind an XYZ sensor (film.rs:524):
# used by the examples and tests; a host that owns an RGBColorSpace passes its own matrix.
SRGB_FROM_XYZ = np.array([[3.2404542, -1.5371385, -0.4985314], [-0.9692660, 1.8760108, 0.0415560], [0.0556434, -0.2040259, 1.0572252]], np.float32)


def film_get_image(lib, film, matrix=None, write_fp16=False):
    """RgbFilm::get_image through the host mirror (shm_film_get_image): (H, W, 3) float32."""
    m = np.ascontiguousarray(np.eye(3, dtype=np.float32) if matrix is None else matrix, dtype=np.float32)
    film = np.ascontiguousarray(film)
    out = np.empty(film.shape + (3,), np.float32)
    abi.check(lib, lib.shm_film_get_image(film.ctypes.data_as(C.c_void_p), film.size, m.ctypes.data_as(abi.c_float_p), int(write_fp16),
                                          out.ctypes.data_as(abi.c_float_p)), "shm_film_get_image")
    return out


def film_to_rgb(film):
    """RgbFilm::get_pixel_rgb without the output colour matrix: rgb_sum / weight_sum as f32 (film.rs:720-731)."""
    rgb = film["rgb_sum"].astype(np.float32)
    w = film["weight_sum"].astype(np.float32)
    out = rgb.copy()
    nz = w != 0
    out[nz] = rgb[nz] / w[nz][..., None]
    return out


def shard_tiles(n_tiles, tiles_per_row, rank, world_size, rows_per_block=None, lib=None):
    """Static interleaved sharding through the C ABI (shm_shard_tiles): blocks of `rows_per_block` tile rows, block b ->
    rank b % world. Default block height: about 16 blocks per rank, so that expensive image regions (the object) and cheap ones
    (walls) are spread over all ranks."""
    lib = lib or abi.load_library()
    idx = (C.c_uint32 * max(1, n_tiles))()
    n = C.c_uint32()
    abi.check(lib, lib.shm_shard_tiles(n_tiles, tiles_per_row, rank, world_size, int(rows_per_block or 0), idx, C.byref(n)), "shm_shard_tiles")
    return np.frombuffer(idx, dtype=np.uint32, count=n.value).astype(np.int64)


class Renderer:
    """Owns a ShmScene on one GPU. Raises if the library or a device is missing: no CPU fallback."""

    def __init__(self, lib, desc, device=0):
        self.lib = lib
        self.desc = desc
        self.handle = C.c_void_p()
        abi.check(lib, lib.shm_scene_create(C.byref(desc), device, C.byref(self.handle)), "shm_scene_create")
        pb = desc.film.pixel_bounds
        self.pixel_bounds = (pb[0], pb[1], pb[2], pb[3])
        self.width, self.height = pb[2] - pb[0], pb[3] - pb[1]
        self.tiles, self.n_tiles = tiles_for(lib, self.pixel_bounds)
        self.tiles_per_row = (self.width + 7) // 8

    def close(self):
        if self.handle:
            self.lib.shm_scene_destroy(self.handle)
            self.handle = C.c_void_p()

    def _tile_subset(self, tile_indices):
        if tile_indices is None:
            return self.tiles, self.n_tiles
        sub = (abi.ShmTile * max(1, len(tile_indices)))()
        for k, i in enumerate(tile_indices):
            sub[k] = self.tiles[int(i)]
        return sub, len(tile_indices)

    def clear(self):
        abi.check(self.lib, self.lib.shm_film_clear(self.handle), "shm_film_clear")

    def render_waves(self, params, tile_indices=None, waves=None):
        """Enqueue all spp-waves into the device film; returns accumulated stats dict."""
        tiles, n = self._tile_subset(tile_indices)
        stats = abi.ShmStats()
        if n == 0:
            return stats.as_dict()
        for (ws, we) in (waves if waves is not None else wave_schedule(params.samples_per_pixel)):
            abi.check(self.lib, self.lib.shm_render_wave(self.handle, C.byref(params), tiles, n, ws, we, C.byref(stats)), "shm_render_wave")
        return stats.as_dict()

    def render_device(self, params, tile_indices=None):
        """Whole render (all spp-waves, fused into <= 64-spp launches) into the device film; returns the stats dict."""
        tiles, n = self._tile_subset(tile_indices)
        stats = abi.ShmStats()
        if n:
            abi.check(self.lib, self.lib.shm_render_device(self.handle, C.byref(params), tiles, n, C.byref(stats)), "shm_render_device")
        return stats.as_dict()

    # ---- multi-GPU behind the C ABI (one process per GPU; include/shimmer_hip.h "multi-GPU") ----
    def dist_unique_id(self):
        """Rank 0: the 128-byte RCCL unique id the host distributes to every rank (any control channel)."""
        buf = (C.c_uint8 * abi.SHM_DIST_ID_BYTES)()
        abi.check(self.lib, self.lib.shm_dist_unique_id(buf), "shm_dist_unique_id")
        return bytes(buf)

    def dist_init(self, rank, world, unique_id):
        buf = (C.c_uint8 * abi.SHM_DIST_ID_BYTES).from_buffer_copy(bytes(unique_id))
        abi.check(self.lib, self.lib.shm_dist_init(self.handle, rank, world, buf), "shm_dist_init")

    def render_sharded(self, params):
        """Collective: clear, render this rank's tile shard, gather the film rows into rank 0's device film over RCCL."""
        stats = abi.ShmStats()
        abi.check(self.lib, self.lib.shm_render_sharded(self.handle, C.byref(params), C.byref(stats)), "shm_render_sharded")
        return stats.as_dict()

    def dist_barrier(self):
        abi.check(self.lib, self.lib.shm_dist_barrier(self.handle), "shm_dist_barrier")

    def dist_allreduce(self, values, op=abi.SHM_REDUCE_SUM):
        """All-reduce of a short list of host doubles over the scene's communicator (identity without one)."""
        buf = (C.c_double * len(values))(*[float(v) for v in values])
        abi.check(self.lib, self.lib.shm_dist_allreduce_f64(self.handle, buf, len(values), op), "shm_dist_allreduce_f64")
        return list(buf)

    def dist_allgather(self, values):
        """Every rank's short list of host doubles, rank-major: [[rank 0's], [rank 1's], ...]."""
        info = self.dist_info()
        n, world = len(values), max(1, info["world"])
        mine = (C.c_double * n)(*[float(v) for v in values])
        out = (C.c_double * (n * world))()
        abi.check(self.lib, self.lib.shm_dist_allgather_f64(self.handle, mine, n, out), "shm_dist_allgather_f64")
        return [list(out[i * n:(i + 1) * n]) for i in range(world)]

    def dist_info(self):
        info = abi.ShmDistInfo()
        abi.check(self.lib, self.lib.shm_dist_info(self.handle, C.byref(info)), "shm_dist_info")
        return info.as_dict()

    def dist_selftest(self):
        abi.check(self.lib, self.lib.shm_dist_selftest(self.handle), "shm_dist_selftest")

    def read_film(self):
        film = np.zeros((self.height, self.width), dtype=FILM_DTYPE)
        abi.check(self.lib, self.lib.shm_film_read(self.handle, film.ctypes.data_as(C.c_void_p)), "shm_film_read")
        return film

    def film_device_ptr(self):
        p, n = C.c_void_p(), C.c_uint64()
        abi.check(self.lib, self.lib.shm_film_device_ptr(self.handle, C.byref(p), C.byref(n)), "shm_film_device_ptr")
        return p.value, n.value

    def render(self, params, tile_indices=None):
        """ImageTileIntegrator::render: clear, all waves, read back. Returns (film structured array, stats)."""
        self.clear()
        stats = self.render_waves(params, tile_indices)
        return self.read_film(), stats

    def trace(self, rays, any_hit=False):
        """Bring-up entry: rays = structured/float32 array (n, 8) [o,d,t_max,pad]. Returns hits (n,8 view) or occluded bytes."""
        rays = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 8)
        n = rays.shape[0]
        stats = abi.ShmStats()
        if any_hit:
            out = np.zeros(n, np.uint8)
            abi.check(self.lib, self.lib.shm_trace_any(self.handle, rays.ctypes.data_as(C.c_void_p), n, out.ctypes.data_as(C.c_void_p), C.byref(stats)), "shm_trace_any")
        else:
            out = np.zeros(n, dtype=HIT_DTYPE)
            abi.check(self.lib, self.lib.shm_trace_closest(self.handle, rays.ctypes.data_as(C.c_void_p), n, out.ctypes.data_as(C.c_void_p), C.byref(stats)), "shm_trace_closest")
        return out, stats.as_dict()


HIT_DTYPE = np.dtype([("prim", "<i4"), ("t", "<f4"), ("b0", "<f4"), ("b1", "<f4"), ("b2", "<f4"), ("phi", "<f4"), ("instance", "<u4"), ("pad", "<u4")])


def film_tensor(renderer, device=None):
    """The renderer's film as a flat float64 torch tensor: zero-copy view of the HBM buffer when `device` is given."""
    import torch

    if device is not None:
        ptr, nbytes = renderer.film_device_ptr()

        class _Holder:
            pass

        h = _Holder()
        h.__cuda_array_interface__ = {"shape": (nbytes // 8,), "typestr": "<f8", "data": (ptr, False), "version": 2}
        return torch.as_tensor(h, device=device)
    return torch.from_numpy(renderer.read_film().view(np.float64).reshape(-1).copy())


def render_multi(lib, desc, devices, params):
    """One process, n devices: shm_render_multi (one host thread + scene replica per device, film rows gathered over xGMI peer copies).
    Returns (film, [stats per device])."""
    pb = desc.film.pixel_bounds
    film = np.zeros((pb[3] - pb[1], pb[2] - pb[0]), dtype=FILM_DTYPE)
    devs = (C.c_int32 * len(devices))(*devices)
    stats = (abi.ShmStats * len(devices))()
    abi.check(lib, lib.shm_render_multi(C.byref(desc), devs, len(devices), C.byref(params), film.ctypes.data_as(C.c_void_p), stats), "shm_render_multi")
    return film, [st.as_dict() for st in stats]


def gather_film(local, rank, world_size, height, width, to_host=True):
    """TEST HARNESS (the product's gather is shm_render_sharded / shm_render_multi behind the C ABI): gather the per-rank films
    (flat float64 tensors) to rank 0 through torch.distributed ('gloo' in the CPU tests, 'nccl' = RCCL on a GPU).  Pixel ownership is exclusive (tiles are disjoint) and untouched
    pixels are exactly 0.0, so adding the slabs reproduces the single-process film bit for bit.
    to_host=False leaves the summed film on rank 0's device (a flat float64 tensor): the read-back is the caller's business
    (the reference writes its image once at the end, integrator.rs:311-321) and stays out of a timed region."""
    import torch
    import torch.distributed as dist

    gathered = [torch.empty_like(local) for _ in range(world_size)] if rank == 0 else None
    dist.gather(local, gathered, dst=0)
    if rank != 0:
        return None
    total = gathered[0].clone()
    for g in gathered[1:]:
        total += g
    if not to_host:
        return total
    return total.cpu().numpy().view(FILM_DTYPE).reshape(height, width)

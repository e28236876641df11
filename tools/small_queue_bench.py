#!/usr/bin/env python3
"""Launch time of the traversal kernel on SMALL queues (late bounces of deep paths, per-rank shards): incoherent rays inside the
crown-proxy's bounds, n from 64 K to 16 M, through shm_trace_closest_device (HIP-event time per launch, 10 launches each).
Env knobs (SHM_TRACE3_BLOCKS_PER_CU, SHM_REFILL_MIN, SHM_LEAF_MIN) are read at scene creation: one process per configuration."""
import ctypes as C
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from shimmer_amd import abi, scenes, render

lib = abi.load_library()
sc = scenes.crown_proxy(lib, 64, 64)
r = render.Renderer(lib, sc.desc, 0)
dev = torch.device("cuda", 0)
rng = np.random.default_rng(7)
nmax = 1 << 24
o = rng.uniform(-2.0, 2.0, (nmax, 3)).astype(np.float32)
d = rng.normal(size=(nmax, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
rays = np.zeros((nmax, 8), np.float32)
rays[:, :3], rays[:, 3:6], rays[:, 6] = o, d, np.inf
t_rays = torch.from_numpy(rays).to(dev)
t_hits = torch.zeros((nmax, 8), dtype=torch.float32, device=dev)
for n in (1 << 16, 1 << 18, 600_000, 1 << 20, 1 << 22, 1 << 24):
    st = abi.ShmStats()
    abi.check(lib, lib.shm_trace_closest_device(r.handle, C.c_void_p(t_rays.data_ptr()), n, C.c_void_p(t_hits.data_ptr()), 2, C.byref(st)), "warm")
    abi.check(lib, lib.shm_trace_closest_device(r.handle, C.c_void_p(t_rays.data_ptr()), n, C.c_void_p(t_hits.data_ptr()), 10, C.byref(st)), "trace")
    ms = st.ms_trace_closest / 10
    print(f"n={n:9d}  {ms:7.3f} ms/launch  {n / ms / 1e3:8.1f} Mray/s  nodes/ray {st.nodes_closest / st.rays_closest:5.1f}", flush=True)

#!/usr/bin/env python3
"""K2 microbenchmark on S3: device-resident ray batches through shm_trace_closest_device.
Ray sets: camera (coherent), diffuse-bounce rays off the object (incoherent), the same sorted by a Morton key.
Prints Mray/s, nodes/ray, algorithmic GB/s. Env knobs (SHM_REFILL_MIN, SHM_LEAF_MIN)
are read at scene creation, so each configuration is a separate process invocation."""
import ctypes as C
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from shimmer_amd import abi, scenes, render

lib = abi.load_library()
n_side = int(os.environ.get("TB_N", "599"))
sc = scenes.ganesha_proxy(lib, 1024, 1024, n=n_side)
r = render.Renderer(lib, sc.desc, 0)
dev = torch.device("cuda", 0)
N = int(os.environ.get("TB_RAYS", str(1 << 22)))
rng = np.random.default_rng(42)

def camera_rays():
    cam = sc.desc.camera
    m = np.array(list(cam.camera_from_raster), np.float64).reshape(4, 4)
    side = int(np.sqrt(N))
    # tile order (8x8) like the renderer
    ys, xs = np.mgrid[0:side, 0:side]
    ty, tx, iy, ix = ys // 8, xs // 8, ys % 8, xs % 8
    order = np.lexsort((iy.ravel(), ix.ravel(), tx.ravel(), ty.ravel()))
    px = (xs.ravel()[order] + 0.5) * (1024.0 / side)
    py = (ys.ravel()[order] + 0.5) * (1024.0 / side)
    p = np.stack([px, py, np.zeros_like(px), np.ones_like(px)], 1) @ m.T
    d = p[:, :3] / p[:, 3:4]
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    # camera space -> render space (PerspectiveCamera::generate_ray, camera.rs:1003-1028: render_from_camera; the pinhole sits at its origin)
    rfc = np.array(list(cam.render_from_camera), np.float64).reshape(4, 4)
    d = d @ rfc[:3, :3].T
    o = np.tile(rfc[:3, 3], (d.shape[0], 1))
    rays = np.zeros((d.shape[0], 8), np.float32)
    rays[:, :3], rays[:, 3:6] = o, d
    rays[:, 6] = np.inf
    return rays

def bounce_rays(first):
    """Second-generation rays: trace `first`, spawn cosine-distributed directions at the hit points."""
    hits = trace(first, 1)[1]
    ok = hits["prim"] >= 0
    o = first[ok, :3] + first[ok, 3:6] * hits["t"][ok, None]
    n = o.shape[0]
    d = rng.normal(size=(n, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    o = o + d * 1e-3
    rays = np.zeros((n, 8), np.float32)
    rays[:, :3], rays[:, 3:6], rays[:, 6] = o, d, np.inf
    return rays

def morton_sort(rays):
    lo, hi = rays[:, :3].min(0), rays[:, :3].max(0)
    q = np.clip(((rays[:, :3] - lo) / np.maximum(hi - lo, 1e-9) * 1023).astype(np.uint64), 0, 1023)
    def spread(v):
        v = (v | (v << 16)) & 0x030000FF
        v = (v | (v << 8)) & 0x0300F00F
        v = (v | (v << 4)) & 0x030C30C3
        v = (v | (v << 2)) & 0x09249249
        return v
    key = spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2)
    octant = ((rays[:, 3] < 0).astype(np.uint64) | ((rays[:, 4] < 0).astype(np.uint64) << 1) | ((rays[:, 5] < 0).astype(np.uint64) << 2))
    key = (octant << 30) | key
    return rays[np.argsort(key, kind="stable")]

def trace(rays, repeat):
    n = rays.shape[0]
    d_rays = torch.from_numpy(rays).to(dev)
    d_hits = torch.empty((n, 8), dtype=torch.float32, device=dev)
    st = abi.ShmStats()
    abi.check(lib, lib.shm_trace_closest_device(r.handle, d_rays.data_ptr(), n, d_hits.data_ptr(), repeat, C.byref(st)), "trace")
    hits = d_hits.cpu().numpy().view(render.HIT_DTYPE).reshape(-1)
    return st, hits

def report(name, rays, repeat=5):
    trace(rays, 1)
    st, hits = trace(rays, repeat)
    n = rays.shape[0]
    ms = st.ms_trace_closest / repeat
    nodes, tris = st.nodes_closest / repeat, st.tris_closest / repeat
    gb = (32 * nodes + 48 * tris + 48 * n) / 1e9
    print(f"{name:28s} n={n:8d} {ms:8.3f} ms  {n/ms/1e3:8.1f} Mray/s  nodes/ray {nodes/n:6.1f} prims/ray {tris/n:5.2f}  alg {gb/ms*1e3:7.0f} GB/s ({gb/ms*1e3/80:5.1f}% of 8 TB/s)  hit {float((hits['prim']>=0).mean()):.2f}", flush=True)

def survey_8d_rays():
    """origins uniform in the scene's bounding sphere x 1.5, directions uniform on the sphere, seed 42 — the incoherent batch."""
    g = np.random.default_rng(42)
    b = sc.info["bounds"]
    lo, hi = b[:, :3].min(0), b[:, 3:].max(0)
    c, rad = (lo + hi) / 2, np.linalg.norm(hi - lo) / 2 * 1.5
    o = g.normal(size=(N, 3)); o /= np.linalg.norm(o, axis=1, keepdims=True)
    o = c + o * (rad * g.random((N, 1)) ** (1.0 / 3.0))
    d = g.normal(size=(N, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = np.zeros((N, 8), np.float32)
    rays[:, :3], rays[:, 3:6], rays[:, 6] = o, d, np.inf
    return rays

cfg = {k: os.environ.get(k) for k in ("SHM_REFILL_MIN", "SHM_LEAF_MIN")}
print("config:", cfg, flush=True)
cam = camera_rays()
report("camera (tile order)", cam)
report("SURVEY 8d incoherent", survey_8d_rays())
b1 = bounce_rays(cam)
report("bounce-1 (path order)", b1)
report("bounce-1 shuffled", b1[rng.permutation(b1.shape[0])])
report("bounce-1 morton-sorted", morton_sort(b1))
b2 = bounce_rays(b1)
report("bounce-2 (path order)", b2)
report("bounce-2 morton-sorted", morton_sort(b2))
r.close()

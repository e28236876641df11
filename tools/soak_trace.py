#!/usr/bin/env python3
"""Soak run of the traversal kernels alone on the headline scene (S3: 4.3 M triangles, 8.5 M nodes): batches of random rays (uniform origins in
1.5 x the scene's bounding sphere, half of them aimed at the object; infinite and finite t_max) through shm_trace_closest / shm_trace_any and
through the oracle — hit records (prim, t, barycentrics) as bits, occlusion flags, node / primitive visit counters.
   python tools/soak_trace.py [batches] [rays per batch] [variant]      variant: patch_emitter | one_sphere | instanced (round 5: the k_trace5<., GEN> kernels — parked
   non-triangle tests, the instance marker; phi and the instance of a hit are compared too)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import oracle_py
from shimmer_amd import abi, render, scenes
batches = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 20
variant = sys.argv[3] if len(sys.argv) > 3 else None
lib = abi.load_library()
sc = scenes.ganesha_proxy(lib, 64, 64, variant=variant)
gpu, orc = render.Renderer(lib, sc.desc, 0), oracle_py.Oracle(sc.desc)
b = sc.info["bounds"]
lo, hi = b[:, :3].min(0), b[:, 3:].max(0)
c, r = (lo + hi) / 2, np.linalg.norm(hi - lo) / 2
bad = 0
t0 = time.time()
for k in range(batches):
    rng = np.random.default_rng(1000 + k)
    o = c + (rng.random((n, 3)) * 2 - 1) * r * 1.5
    d = rng.normal(size=(n, 3))
    aim = rng.random(n) < 0.5
    d[aim] = (c + (rng.random((int(aim.sum()), 3)) - 0.5) * r * 0.5) - o[aim]
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = np.zeros((n, 8), np.float32)
    rays[:, :3], rays[:, 3:6] = o, d
    rays[:, 6] = np.inf if k % 2 == 0 else rng.uniform(0.5, 6.0, n).astype(np.float32)
    hg, sg = gpu.trace(rays)
    ho, so = orc.trace(rays)
    hit = ho["prim"] >= 0
    same = np.array_equal(hg["prim"], ho["prim"]) and all(np.array_equal(hg[f].view(np.uint32)[hit], ho[f].view(np.uint32)[hit]) for f in ("t", "b0", "b1", "b2", "phi", "instance"))
    same = same and sg["nodes_closest"] == so["nodes_closest"] and sg["tris_closest"] == so["tris_closest"]
    ag, s2 = gpu.trace(rays, any_hit=True)
    ao, s3 = orc.trace(rays, any_hit=True)
    same = same and np.array_equal(ag, ao) and s2["nodes_any"] == s3["nodes_any"] and s2["tris_any"] == s3["tris_any"]
    bad += 0 if same else 1
    print(f"batch {k}: {n} rays, hits {int(hit.sum())}, occluded {int(ao.sum())}, nodes/ray {so['nodes_closest'] / n:.1f}: {'bit-equal' if same else 'MISMATCH'}", flush=True)
print(f"{variant or 'S3'}: {batches} batches x {n} rays: {batches - bad} bit-equal, {bad} mismatching; {time.time() - t0:.0f} s", flush=True)
gpu.close(); orc.close()

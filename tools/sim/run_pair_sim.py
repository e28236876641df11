#!/usr/bin/env python3
"""CPU: pair-step traversal simulation on S3 (camera rays + diffuse bounce rays). python tools/sim/run_pair_sim.py [n_side] [res]"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
from shimmer_amd import abi, scenes
import oracle_py

n_side = int(sys.argv[1]) if len(sys.argv) > 1 else 599
res = int(sys.argv[2]) if len(sys.argv) > 2 else 256
lib = abi.load_library()
sc = scenes.ganesha_proxy(lib, 1024, 1024, n=n_side)
orc = oracle_py.Oracle(sc.desc)
sim = C.CDLL(os.path.join(ROOT, "tools/sim/_build/libpairsim.so"))

class PairStats(C.Structure):
    _fields_ = [(k, C.c_uint64) for k in ("rays", "visits", "pair_steps", "pushes", "pops_pass", "pops_culled", "direct_far", "leaf_visits", "root_leaf")] + \
               [("depth_hist", C.c_uint64 * 66), ("push_depth_hist", C.c_uint64 * 66), ("mismatches", C.c_uint64)]

# the sim library has its own copy of the oracle: create the scene there
sim.orc_scene_create.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
h = C.c_void_p()
assert sim.orc_scene_create(C.byref(sc.desc), C.byref(h)) == 0

def camera_rays(side):
    out = (C.c_float * 14)()
    rays = np.zeros((side * side, 8), np.float32)
    step = 1024 // side
    k = 0
    for y in range(0, 1024, step):
        for x in range(0, 1024, step):
            orc.lib.orc_fn_camera_ray(orc.handle, x, y, 0, 0, out)
            rays[k, :6] = out[:6]; rays[k, 6] = np.inf; k += 1
    return rays

def bounce(first, rng):
    hits, _ = orc.trace(first)
    ok = hits["prim"] >= 0
    o = first[ok, :3] + first[ok, 3:6] * hits["t"][ok, None]
    d = rng.normal(size=(o.shape[0], 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = np.zeros((o.shape[0], 8), np.float32)
    rays[:, :3], rays[:, 3:6], rays[:, 6] = o + d * 1e-3, d, np.inf
    return rays

def report(name, rays):
    st = PairStats()
    t0 = time.time()
    sim.orc_sim_pair(h, rays.ctypes.data_as(C.c_void_p), rays.shape[0], C.byref(st))
    n = st.rays
    print(f"{name}: {n} rays ({time.time()-t0:.1f}s) mismatches {st.mismatches}")
    print(f"  visits/ray {st.visits/n:.2f}  pair_steps/ray {st.pair_steps/n:.2f}  leaf_visits/ray {st.leaf_visits/n:.2f}  pushes/ray {st.pushes/n:.2f}  "
          f"pops pass/ray {st.pops_pass/n:.2f} culled/ray {st.pops_culled/n:.2f}  direct far/ray {st.direct_far/n:.2f}")
    dh = np.array(list(st.depth_hist), np.float64); ph = np.array(list(st.push_depth_hist), np.float64)
    cd = np.cumsum(dh) / dh.sum()
    print("  max stack depth per ray: " + " ".join(f"{i}:{cd[i]:.3f}" for i in range(0, 24, 2)))
    cp = np.cumsum(ph) / max(ph.sum(), 1)
    print("  pushes landing at level < L: " + " ".join(f"{i+1}:{cp[i]:.4f}" for i in range(3, 20, 2)))

rng = np.random.default_rng(42)
cam = camera_rays(res)
report("camera", cam)
b1 = bounce(cam, rng)
report("bounce1", b1)
b2 = bounce(b1, rng)
report("bounce2", b2)

import sys, time, ctypes as C
sys.path.insert(0,'/root/repo')
import numpy as np
from shimmer_amd import abi, scenes
lib = abi.load_library()
t0=time.perf_counter()
verts, tris = scenes.cube_sphere(599)
t1=time.perf_counter()
print("cube_sphere", t1-t0)
p = verts[tris]  # (n,3,3)
b = np.concatenate([p.min(1), p.max(1)], 1).astype(np.float32)
n = b.shape[0]
nodes = (abi.ShmBvhNode * (2*n))()
cnt = C.c_uint32()
order = np.zeros(n, np.uint32)
t2=time.perf_counter()
rc = lib.shm_bvh_build(b.ctypes.data_as(C.POINTER(C.c_float)), n, 0, nodes, C.byref(cnt), order.ctypes.data_as(C.POINTER(C.c_uint32)))
t3=time.perf_counter()
print("rc", rc, "nodes", cnt.value, "build s", t3-t2)
import hashlib
print(hashlib.sha256(bytes(memoryview(nodes))[:cnt.value*32]).hexdigest()[:16], hashlib.sha256(order.tobytes()).hexdigest()[:16])

# C4 frame times, frame by frame, in a fresh process and after other scenes' renderers have come and gone (the bench line's side results showed a slower second frame)
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from shimmer_amd import abi, scenes, render
lib = abi.load_library()
def frames(sc, spp, depth, n=5):
    r = render.Renderer(lib, sc.desc, 0)
    p = render.make_params(seed=0, spp=spp, max_depth=depth)
    r.clear(); r.render_device(p)
    out = []
    for _ in range(n):
        r.clear()
        t0 = time.perf_counter(); st = r.render_device(p); dt = time.perf_counter() - t0
        out.append((round(dt * 1e3, 1), round(st["ms_trace_closest"], 1), round(st["ms_trace_any"], 1), round(st["ms_shade"], 1)))
    r.close()
    return out
c4 = scenes.crown_proxy(lib, 1000, 1400)
print("C4 fresh process   (frame ms, closest, any, shade):", frames(c4, 256, 32), flush=True)
for k in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
    s = scenes.ganesha_proxy(lib, 512, 512, n=100, variant=[None, "one_sphere", "instanced", "quads", "environment", "textured_floor"][k % 6])
    frames(s, 16, 5, n=1)
    del s
print("C4 after renderers (frame ms, closest, any, shade):", frames(c4, 256, 32), flush=True)
print("C4 again                                          :", frames(c4, 256, 32), flush=True)

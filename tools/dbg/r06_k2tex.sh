cd /root/repo
python3 - <<'PY'
import sys, time
sys.path.insert(0,'.')
from shimmer_amd import abi, scenes, render
lib=abi.load_library()
for name, kw in (("S3", {}), ("S3n", dict(variant="smooth")), ("S3to", dict(variant="textured_object")), ("S3t", dict(variant="textured_floor"))):
    sc=scenes.ganesha_proxy(lib, 1024, 1024, **kw)
    r=render.Renderer(lib, sc.desc, 0)
    p=render.make_params(seed=0, spp=64, max_depth=5)
    r.clear(); r.render_device(p); r.clear()
    st=r.render_device(p)
    print(f"{name:5s} rays_closest {st['rays_closest']/1e6:7.1f} M nodes/ray {st['nodes_closest']/st['rays_closest']:6.2f} prims/ray {st['tris_closest']/st['rays_closest']:5.2f} | K2 {st['ms_trace_closest']:6.1f} ms = {st['rays_closest']/st['ms_trace_closest']/1e3:7.1f} Mray/s | rays_any {st['rays_any']/1e6:7.1f} M nodes/ray {st['nodes_any']/max(1,st['rays_any']):6.2f} K3 {st['ms_trace_any']:6.1f} ms | launches {st['launches_closest']}", flush=True)
    r.close()
PY

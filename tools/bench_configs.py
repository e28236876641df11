#!/usr/bin/env python3
"""Throughput of the BASELINE.json side configurations on one GPU (informational; bench.py measures the headline one):
C1 sphere + area light 128^2 x 4, C2 Cornell 512^2 x 64, C4 crown-proxy 1000x1400 x 256 maxdepth 32, and the coated S3."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from shimmer_amd import abi, scenes, render

lib = abi.load_library()
CONFIGS = [
    ("C1 sphere+light 128x128x4", lambda: scenes.sphere_light(lib, 128, 128), 4, 5),
    ("C2 cornell 512x512x64", lambda: scenes.cornell_box(lib, 512, 512), 64, 5),
    ("C2p cornell+patches 512x512x64", lambda: scenes.cornell_box(lib, 512, 512, patches=True), 64, 5),
    ("C4 crown-proxy 1000x1400x256 d32", lambda: scenes.crown_proxy(lib, 1000, 1400), 256, 32),
    ("S3c coated ganesha 1024x1024x64", lambda: scenes.ganesha_proxy(lib, 1024, 1024, coated=True), 64, 5),
    ("S3 headline ganesha 1024x1024x256", lambda: scenes.ganesha_proxy(lib, 1024, 1024), 256, 5),
    ("S3p ganesha, patch emitter 1024x1024x256", lambda: scenes.ganesha_proxy(lib, 1024, 1024, variant="patch_emitter"), 256, 5),
    ("S3s ganesha + one sphere 1024x1024x256", lambda: scenes.ganesha_proxy(lib, 1024, 1024, variant="one_sphere"), 256, 5),
    ("S3i ganesha instanced 1024x1024x256", lambda: scenes.ganesha_proxy(lib, 1024, 1024, variant="instanced"), 256, 5),
    ("S3ig ganesha as 4 x 4 x 4 instances of one small definition 1024x1024x256", lambda: scenes.ganesha_proxy(lib, 1024, 1024, variant="instance_grid"), 256, 5),
    ("S3q ganesha as 2.15 M bilinear patches (a quad PLY) 1024x1024x256", lambda: scenes.ganesha_proxy(lib, 1024, 1024, variant="quads"), 256, 5),
    ("S3e ganesha under an environment map 1024x1024x256", lambda: scenes.ganesha_proxy(lib, 1024, 1024, variant="environment"), 256, 5),
    ("S3to ganesha, textured OBJECT (image texture over its uv) 1024x1024x256", lambda: scenes.ganesha_proxy(lib, 1024, 1024, variant="textured_object"), 256, 5),
    ("S3n ganesha, per-vertex normals + uv 1024x1024x256", lambda: scenes.ganesha_proxy(lib, 1024, 1024, variant="smooth"), 256, 5),
    ("S3m ganesha, 8192-triangle emitter 1024x1024x256", lambda: scenes.ganesha_proxy(lib, 1024, 1024, variant="mesh_emitter"), 256, 5),
    ("S3t ganesha, textured floor 1024x1024x256", lambda: scenes.ganesha_proxy(lib, 1024, 1024, variant="textured_floor"), 256, 5),
    ("S3tb ganesha, textured floor (bilinear) 1024x1024x256", lambda: scenes.ganesha_proxy(lib, 1024, 1024, variant="textured_floor", floor_filter="bilinear"), 256, 5),
    ("S3tp ganesha, textured floor (point) 1024x1024x256", lambda: scenes.ganesha_proxy(lib, 1024, 1024, variant="textured_floor", floor_filter="point"), 256, 5),
    ("S3th ganesha, a textured material out of sight 1024x1024x256", lambda: scenes.ganesha_proxy(lib, 1024, 1024, variant="textured_hidden"), 256, 5),
    ("S3c256 coated ganesha 1024x1024x256", lambda: scenes.ganesha_proxy(lib, 1024, 1024, coated=True), 256, 5),
    ("S3ce coated ganesha under an environment map 1024x1024x256", lambda: scenes.ganesha_proxy(lib, 1024, 1024, coated=True, variant="environment"), 256, 5),
    ("C4e crown-proxy under an environment map 1000x1400x256 d32", lambda: scenes.crown_proxy(lib, 1000, 1400, environment=scenes.environment_image(64)), 256, 32),
    ("C2t cornell textured 512x512x64", lambda: scenes.cornell_box(lib, 512, 512, textured=True), 64, 6),
    ("C2u cornell textured, no coated 512x512x64", lambda: scenes.cornell_box(lib, 512, 512, textured=True, textured_coated_ceiling=False), 64, 6),
    ("C2f-point cornell textured, every filter point", lambda: scenes.cornell_box(lib, 512, 512, textured=True, texture_filter="point"), 64, 6),
    ("C2f-bilinear cornell textured, every filter bilinear", lambda: scenes.cornell_box(lib, 512, 512, textured=True, texture_filter="bilinear"), 64, 6),
    ("C2f-trilinear cornell textured, every filter trilinear", lambda: scenes.cornell_box(lib, 512, 512, textured=True, texture_filter="trilinear"), 64, 6),
    ("C2f-ewa cornell textured, every filter ewa", lambda: scenes.cornell_box(lib, 512, 512, textured=True, texture_filter="ewa"), 64, 6),
    ("E3 spheres + environment map 512x384x64", lambda: scenes.three_spheres(lib, 512, 384, camera=(0.75, 0.5, 9.0), environment=scenes.environment_image(64)), 64, 5),
]
only = sys.argv[1:]
KEYS = {c[0].split()[0] for c in CONFIGS}
for name, make, spp, depth in CONFIGS:
    # an argument that is a configuration's key ("S3", "S3c") selects that one; anything else is a substring of the name
    if only and not any(name.split()[0] == o if o in KEYS else o in name for o in only):
        continue
    sc = make()
    r = render.Renderer(lib, sc.desc, 0)
    p = render.make_params(seed=0, spp=spp, max_depth=depth)
    r.clear(); r.render_device(p)  # warm-up (workspace allocation)
    r.clear()
    t0 = time.perf_counter()
    st = r.render_device(p)
    dt = time.perf_counter() - t0
    rays = st["rays_closest"] + st["rays_any"]
    print(f"{name:36s} prims {sc.info['n_primitives']:8d}  {rays/1e6:9.1f} Mrays in {dt*1e3:8.1f} ms = {rays/dt/1e6:8.1f} Mray/s | closest {st['ms_trace_closest']:7.1f} any {st['ms_trace_any']:7.1f} "
          f"shade {st['ms_shade']:7.1f} ms | nodes/ray {st['nodes_closest']/max(1,st['rays_closest']):5.1f}", flush=True)
    r.close()

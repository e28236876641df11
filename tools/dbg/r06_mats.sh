cd /root/repo
python3 - <<'PY'
import sys, os
sys.path.insert(0,'.'); sys.path.insert(0,'oracle')
import numpy as np
from shimmer_amd import abi, scenes, render
import oracle_py
lib=abi.load_library()
for m in ("gold","glass","coated_conductor"):
    sc=scenes.ganesha_proxy(lib, 64, 64, n=32, object_material=m)
    g=render.Renderer(lib, sc.desc, 0); o=oracle_py.Oracle(sc.desc)
    p=render.make_params(seed=3, spp=6, max_depth=8)
    fg,sg=g.render(p); fo,so=o.render(p, n_threads=os.cpu_count())
    print(m, "bit-exact", np.array_equal(fg,fo), all(sg[k]==so[k] for k in ("paths","rays_closest","rays_any","nodes_closest","tris_closest","nodes_any","tris_any")))
    g.close(); o.close()
PY
python3 tools/film_ab.py --scenes S3au,S3gl,S3gl16,S3cc --rounds 2 "" 2>&1 | grep -v "^$"

cd /root/repo
export TMPDIR=/tmp
for V in smooth textured_object; do
cat > /tmp/run_$V.py <<PY
import sys
sys.path.insert(0,'/root/repo')
from shimmer_amd import abi, scenes, render
lib=abi.load_library()
sc=scenes.ganesha_proxy(lib, 1024, 1024, variant="$V")
r=render.Renderer(lib, sc.desc, 0)
p=render.make_params(seed=0, spp=256, max_depth=5)
r.clear(); r.render_device(p); r.clear()
st=r.render_device(p)
print("$V", st['rays_closest'], st['ms_trace_closest'], st['ms_trace_any'], st['ms_shade'])
r.close()
PY
echo "== $V: per-bounce queue sizes (SHM_DEBUG; second frame)"
SHM_DEBUG=1 python3 /tmp/run_$V.py 2>&1 | grep -E "bounce [0-9]+:" | tail -6
echo "== $V: per-dispatch K2 / K3 durations (rocprofv3 --kernel-trace; both frames)"
rocprofv3 --kernel-trace -d gpurun_out/prof_r06_k2_$V -o kt -- python3 /tmp/run_$V.py > /dev/null 2>&1
python3 - <<PY
import sqlite3, glob
db=glob.glob('gpurun_out/prof_r06_k2_$V/**/*.db', recursive=True)[0]
c=sqlite3.connect(db).cursor()
rows=list(c.execute("select S.display_name, K.start, K.end from rocpd_kernel_dispatch K join rocpd_info_kernel_symbol S on S.id=K.kernel_id and S.guid=K.guid order by K.start"))
for want in ("k_trace5<false","k_trace5<true"):
    d=[(e-s)/1e6 for n,s,e in rows if want in n]
    print(want, " ".join(f"{x:6.2f}" for x in d[-6:]), "| sum", f"{sum(d[-6:]):.1f}")
PY
done

export TMPDIR=/tmp
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_TC_INST_REQ -d gpurun_out/icache -o pmc -- python3 bench.py --pmc-child --steps 1 --warmup 0 --spp 256 > gpurun_out/icache.log 2>&1
python3 - <<'PY'
import sqlite3, glob, re, collections
db=glob.glob('gpurun_out/icache/**/*.db', recursive=True)[0]
cur=sqlite3.connect(db).cursor()
acc=collections.defaultdict(lambda: collections.defaultdict(float))
for name,c,v in cur.execute("select kernel_name, counter_name, value from counters_collection"):
    k=re.sub(r"\(.*","",name)[:60]
    acc[k][c]+=v
for k,d in acc.items():
    if d.get('SQC_ICACHE_REQ',0)>1e6: print(k, {c:f"{v:.3e}" for c,v in d.items()}, "miss rate %.3f"%(d['SQC_ICACHE_MISSES']/max(1,d['SQC_ICACHE_REQ'])))
PY

cd /root/repo
export TMPDIR=/tmp
for C in C2t S3t; do
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_TC_INST_REQ -d gpurun_out/icache_$C -o pmc -- python3 tools/bench_configs.py $C > gpurun_out/icache_$C.log 2>&1
python3 - $C <<'PY'
import sqlite3, glob, re, collections, sys
db=glob.glob('gpurun_out/icache_%s/**/*.db'%sys.argv[1], recursive=True)[0]
cur=sqlite3.connect(db).cursor()
acc=collections.defaultdict(lambda: collections.defaultdict(float))
for name,c,v in cur.execute("select kernel_name, counter_name, value from counters_collection"):
    k=re.sub(r"\(anonymous namespace\)::","",name)
    k=re.sub(r"\(.*","",k)[:60]
    acc[k][c]+=v
print("==", sys.argv[1])
for k,d in acc.items():
    if d.get('SQC_ICACHE_REQ',0)>1e6: print(k, {c:f"{v:.3e}" for c,v in d.items()}, "miss rate %.4f"%(d['SQC_ICACHE_MISSES']/max(1,d['SQC_ICACHE_REQ'])))
PY
done

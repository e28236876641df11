cd /root/repo
python tools/bench_configs.py S3tb 2>&1 | tail -1
SHM_FUSED_TEX=2 python tools/bench_configs.py S3tb 2>&1 | tail -1

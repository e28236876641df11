cd /root/repo
python tools/bench_configs.py S3t 2>&1 | tail -1

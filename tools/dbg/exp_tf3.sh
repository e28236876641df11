cd /root/repo
python tools/bench_configs.py S3t S3tb S3tp 2>&1 | tail -3

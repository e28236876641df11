cd /root/repo
python tools/bench_configs.py S3ce 2>&1 | tail -1
python tools/bench_configs.py "S3c " 2>&1 | tail -1

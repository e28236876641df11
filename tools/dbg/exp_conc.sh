cd /root/repo
for i in 1 2; do
echo "== default"; python tools/bench_configs.py S3c256 S3ce 2>&1 | tail -2 | cut -c1-170
echo "== SHM_CONCURRENT_BIG=0"; SHM_CONCURRENT_BIG=0 python tools/bench_configs.py S3c256 S3ce 2>&1 | tail -2 | cut -c1-170
done

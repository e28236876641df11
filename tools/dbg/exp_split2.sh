cd /root/repo
timeout 2400 python -m pytest tests/test_gpu_parity.py -x -q > gpurun_out/r05_gputests_j.log 2>&1; grep -E "passed|failed|rror" gpurun_out/r05_gputests_j.log | tail -5 | cut -c1-800
for i in 1 2; do
echo "== split"; python tools/bench_configs.py S3c256 S3ce 2>&1 | tail -2
echo "== no split"; SHM_SPLIT_PASS=0 python tools/bench_configs.py S3c256 S3ce 2>&1 | tail -2
done

cd /root/repo
timeout 1800 python -m pytest tests/test_gpu_parity.py -x -q > gpurun_out/r05_gputests_g.log 2>&1; grep -E "passed|failed|rror" gpurun_out/r05_gputests_g.log | tail -5 | cut -c1-800
for i in 1 2; do
echo "== env"; python tools/bench_configs.py S3ce 2>&1 | tail -1
echo "== textured class"; SHM_ENV_LEAN=0 python tools/bench_configs.py S3ce 2>&1 | tail -1
done

cd /root/repo
timeout 1800 python -m pytest tests/test_gpu_parity.py -x -q > gpurun_out/r05_gputests_f.log 2>&1; grep -E "passed|failed|rror" gpurun_out/r05_gputests_f.log | tail -5 | cut -c1-600
for i in 1 2; do
echo "== env plain"; python tools/bench_configs.py C4e 2>&1 | tail -1
echo "== textured class"; SHM_ENV_LEAN=0 python tools/bench_configs.py C4e 2>&1 | tail -1
done

cd /root/repo
timeout 2400 python -m pytest tests/test_gpu_parity.py -x -q > gpurun_out/r05_gputests_i.log 2>&1; grep -E "passed|failed|rror" gpurun_out/r05_gputests_i.log | tail -5 | cut -c1-800
for i in 1 2; do
echo "== split"; python tools/bench_configs.py S3t S3th C2t 2>&1 | tail -3
echo "== no split"; SHM_TEX_SPLIT=0 python tools/bench_configs.py S3t S3th 2>&1 | tail -2
done
SHM_TEX_SPLIT=1 python tools/bench_configs.py C2t 2>&1 | tail -1

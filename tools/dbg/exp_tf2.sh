cd /root/repo
timeout 2400 python -m pytest tests/test_gpu_parity.py -x -q > gpurun_out/r05_gputests_h.log 2>&1; grep -E "passed|failed|rror" gpurun_out/r05_gputests_h.log | tail -5 | cut -c1-800
for i in 1 2; do
echo "== divert"; python tools/bench_configs.py S3t C2t C2u 2>&1 | tail -3
echo "== no divert"; SHM_LEAN_DIVERT=0 python tools/bench_configs.py S3t C2t C2u 2>&1 | tail -3
done

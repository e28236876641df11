cd /root/repo
timeout 1800 python -m pytest tests/test_gpu_parity.py -x -q > gpurun_out/r05_gputests_c.log 2>&1; grep -E "passed|failed|rror" gpurun_out/r05_gputests_c.log | tail -3 | cut -c1-300
for i in 1 2; do
echo "== fused"; python tools/bench_configs.py E3 2>&1 | tail -1
echo "== staged"; SHM_FUSED_GEN=0 python tools/bench_configs.py E3 2>&1 | tail -1
done

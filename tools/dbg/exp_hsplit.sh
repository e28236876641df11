cd /root/repo
timeout 1800 python -m pytest tests/test_gpu_parity.py -x -q > gpurun_out/r05_gputests_n.log 2>&1; grep -E "passed|failed|rror" gpurun_out/r05_gputests_n.log | tail -3 | cut -c1-500
for i in 1 2; do
echo "== split records"; python tools/bench_configs.py S3p S3s C2p E3 2>&1 | tail -4
echo "== 32-byte"; SHM_HIT16=0 python tools/bench_configs.py S3p S3s C2p E3 2>&1 | tail -4
done

cd /root/repo
timeout 1800 python -m pytest tests/test_gpu_parity.py -x -q > gpurun_out/r05_gputests_l.log 2>&1; grep -E "passed|failed|rror" gpurun_out/r05_gputests_l.log | tail -3 | cut -c1-400
for i in 1 2; do
echo "== hit16 in textured scenes"; python tools/bench_configs.py S3t C2t C2u S3ce 2>&1 | tail -4
echo "== off"; SHM_HIT16=0 python tools/bench_configs.py S3t C2t C2u S3ce 2>&1 | tail -4
done

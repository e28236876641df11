cd /root/repo
B=shimmer_amd/csrc/_exp/lib_before_mat8.so
timeout 1800 python -m pytest tests/test_gpu_parity.py -x -q > gpurun_out/r05_gputests_k.log 2>&1; grep -E "passed|failed|rror" gpurun_out/r05_gputests_k.log | tail -3 | cut -c1-400
for i in 1 2; do
echo "== mat8"; python tools/bench_configs.py C4 S3t S3c256 C2t 2>&1 | tail -4
echo "== before"; SHM_LIB=$B python tools/bench_configs.py C4 S3t S3c256 C2t 2>&1 | tail -4
done

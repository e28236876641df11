cd /root/repo
L=shimmer_amd/csrc/_exp/lib_k_trace_K5_QUEUE_PREFETCH_1.so
SHM_LIB=$L timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "trace or render_parity or mixed_shape or c4_frame" 2>&1 | grep -E "passed|failed" | tail -2
for i in 1 2; do
echo "== base"; python tools/bench_configs.py "S3 headline" C4 2>&1 | tail -2
echo "== queue prefetch"; SHM_LIB=$L python tools/bench_configs.py "S3 headline" C4 2>&1 | tail -2
done

cd /root/repo
L=shimmer_amd/csrc/_exp/lib_k_trace_K5_GEN_CLOSEST_WAVES_8.so
SHM_LIB=$L timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "tree_shapes or mixed_shape or S3_small" 2>&1 | grep -E "passed|failed" | tail -1
for i in 1 2; do
echo "== 7 waves"; python tools/bench_configs.py S3p S3s S3i 2>&1 | tail -3
echo "== 8 waves"; SHM_LIB=$L python tools/bench_configs.py S3p S3s S3i 2>&1 | tail -3
done

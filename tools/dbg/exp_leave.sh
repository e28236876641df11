cd /root/repo
L=shimmer_amd/csrc/_exp/lib_k_trace_K5_LEAVE_INLINE_1.so
SHM_LIB=$L timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "instanc or tree_shapes or mixed_shape" 2>&1 | grep -E "passed|failed" | tail -2
for i in 1 2; do
echo "== parked leave"; python tools/bench_configs.py S3i 2>&1 | tail -1
echo "== inline leave"; SHM_LIB=$L python tools/bench_configs.py S3i 2>&1 | tail -1
done

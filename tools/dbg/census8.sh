cd /root/repo
for K in 1 2; do for C in "S3 headline" S3s S3i C4; do
echo "=== census $K : $C"; SHM_LIB=shimmer_amd/csrc/_exp/lib_k_trace_K5_CENSUS_$K.so python tools/bench_configs.py "$C" 2>&1 | grep -v amdgpu | tail -6
done; done

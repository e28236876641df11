cd /root/repo
for v in S3s S3i; do
echo "== $v census closest"; SHM_LIB=shimmer_amd/csrc/_exp/lib_k_trace_K5_CENSUS_1.so python tools/bench_configs.py "$v " 2>&1 | tail -5
done

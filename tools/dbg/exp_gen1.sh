cd /root/repo
echo "== headline default"; python tools/bench_configs.py "S3 headline" 2>&1 | tail -1
echo "== headline HIT16=0"; SHM_HIT16=0 python tools/bench_configs.py "S3 headline" 2>&1 | tail -1
echo "== headline GEN kernels (HIT16=0)"; SHM_TRACE_GEN=1 SHM_HIT16=0 python tools/bench_configs.py "S3 headline" 2>&1 | tail -1
for v in S3s S3i S3p; do
echo "== $v census closest"; SHM_LIB=shimmer_amd/csrc/_exp/lib_k_trace_K5_CENSUS_1.so python tools/bench_configs.py "$v " 2>&1 | tail -5
echo "== $v census any"; SHM_LIB=shimmer_amd/csrc/_exp/lib_k_trace_K5_CENSUS_2.so python tools/bench_configs.py "$v " 2>&1 | tail -5
done

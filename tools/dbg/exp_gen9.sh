cd /root/repo
for i in 1 2; do
echo "== 7 waves"; python tools/bench_configs.py "S3s " "S3p " 2>&1 | tail -2
echo "== 8 waves (closest)"; SHM_LIB=shimmer_amd/csrc/_exp/lib_k_trace_K5_GEN_CLOSEST_WAVES_8.so python tools/bench_configs.py "S3s " "S3p " 2>&1 | tail -2
done

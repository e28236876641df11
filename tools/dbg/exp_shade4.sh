cd /root/repo
for i in 1 2; do
echo "== 3 waves"; python tools/bench_configs.py "S3 headline" 2>&1 | tail -1
echo "== 4 waves"; SHM_LIB=shimmer_amd/csrc/_exp/lib_k_shade_lean_K_SHADE_LEAN_WAVES_4.so python tools/bench_configs.py "S3 headline" 2>&1 | tail -1
done

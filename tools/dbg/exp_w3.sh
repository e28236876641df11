cd /root/repo
for i in 1 2; do
echo "== 2 waves"; python tools/bench_configs.py C4 2>&1 | tail -1
echo "== 3 waves"; SHM_LIB=shimmer_amd/csrc/_exp/lib_k_shade_tail_sorted_K_SHADE_LEAN_WAVES_3.so python tools/bench_configs.py C4 2>&1 | tail -1
done

cd /root/repo
E=shimmer_amd/csrc/_exp
for i in 1 2; do
echo "== tail 3 waves"; SHM_LIB=$E/lib_k_shade_tail_sorted_K_SHADE_LEAN_WAVES_3.so python tools/bench_configs.py C4 2>&1 | tail -1
echo "== tail 4 waves"; SHM_LIB=$E/lib_k_shade_tail_sorted_K_SHADE_LEAN_WAVES_4.so python tools/bench_configs.py C4 2>&1 | tail -1
echo "== tex 2 waves"; python tools/bench_configs.py C2u 2>&1 | tail -1
echo "== tex 3 waves"; SHM_LIB=$E/lib_k_shade_fused_tex_K_SHADE_LEAN_WAVES_3.so python tools/bench_configs.py C2u 2>&1 | tail -1
echo "== gen tex 2 waves"; python tools/bench_configs.py E3 2>&1 | tail -1
echo "== gen tex 3 waves"; SHM_LIB=$E/lib_k_shade_fused_gen_tex_K_SHADE_LEAN_WAVES_3.so python tools/bench_configs.py E3 2>&1 | tail -1
done

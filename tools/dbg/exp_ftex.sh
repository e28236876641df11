cd /root/repo
timeout 1800 python -m pytest tests/test_gpu_parity.py tests/test_gpu_multi.py -x -q > gpurun_out/r05_gputests_b.log 2>&1; tail -3 gpurun_out/r05_gputests_b.log | cut -c1-300
for i in 1 2; do
echo "== fused tex"; python tools/bench_configs.py C2u 2>&1 | tail -1
echo "== staged"; SHM_FUSED_TEX=0 python tools/bench_configs.py C2u 2>&1 | tail -1
done
python tools/bench_configs.py C2f 2>&1 | grep -v amdgpu | tail -4
SHM_FUSED_TEX=0 python tools/bench_configs.py C2f 2>&1 | grep -v amdgpu | tail -4

cd /root/repo
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "trace_bitwise or both_step or edge_cases" 2>&1 | tail -3
echo "== headline GEN kernels (HIT16=0)"; SHM_TRACE_GEN=1 SHM_HIT16=0 python tools/bench_configs.py "S3 headline" 2>&1 | tail -1
python tools/bench_configs.py "S3s " "S3i " "S3p " 2>&1 | tail -3

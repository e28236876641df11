cd /root/repo
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "trace_bitwise or both_step or edge_cases" 2>&1 | tail -2
python tools/bench_configs.py "S3s " "S3i " "S3p " 2>&1 | tail -3
for om in 4 8 24; do echo "== SHM_OTHER_MIN=$om"; SHM_OTHER_MIN=$om python tools/bench_configs.py "S3s " "S3i " "S3p " 2>&1 | tail -3; done

cd /root/repo
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "trace_bitwise or tree_shapes or edge_cases or render_parity" 2>&1 | tail -2
python tools/bench_configs.py "S3s " "S3i " "S3p " "C2p" 2>&1 | tail -4

cd /root/repo
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "render_parity or trace_bitwise" 2>&1 | tail -2
python tools/bench_configs.py "S3p " "C2p" 2>&1 | tail -2

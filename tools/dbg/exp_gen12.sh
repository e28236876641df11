cd /root/repo
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "mixed_shape_frames" 2>&1 | tail -3
python tools/bench_configs.py "S3s " 2>&1 | tail -1

cd /root/repo
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -k "environment" > gpurun_out/r05_env_tests.log 2>&1; grep -E "passed|failed|rror" gpurun_out/r05_env_tests.log | tail -3 | cut -c1-400
python tools/bench_configs.py S3e 2>&1 | tail -1
SHM_FUSED_TEX=0 python tools/bench_configs.py S3e 2>&1 | tail -1

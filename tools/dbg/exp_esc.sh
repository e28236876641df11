cd /root/repo
timeout 1800 python -m pytest tests/test_gpu_parity.py -x -q -k "environment or S3_small or ab_switches or three_spheres" > gpurun_out/r05_gputests_m.log 2>&1; grep -E "passed|failed|rror" gpurun_out/r05_gputests_m.log | tail -3 | cut -c1-400
for i in 1 2; do python tools/bench_configs.py S3e E3 2>&1 | tail -2; done

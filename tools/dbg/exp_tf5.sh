cd /root/repo
python tools/bench_configs.py S3th 2>&1 | tail -1
tools/profile_side.sh r05 S3th > /dev/null 2>&1; grep -E "calls=" gpurun_out/prof_r05_S3th/summary.txt | head -8

cd /root/repo
( time timeout 3000 python -m pytest tests -x -q -m gpu ) > gpurun_out/r05_gputests_full.log 2>&1; grep -E "passed|failed|rror|real" gpurun_out/r05_gputests_full.log | tail -4 | cut -c1-300
python3 tools/kernel_resources.py > gpurun_out/r05_kernel_resources.txt 2>&1
tools/profile_side.sh r05 C2t C2u S3e S3ce C4e > gpurun_out/r05_prof_side6.log 2>&1
bash tools/dbg/bench_final.sh

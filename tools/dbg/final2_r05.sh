cd /root/repo
( time timeout 3000 python -m pytest tests -x -q -m gpu ) > gpurun_out/r05_gputests_full.log 2>&1; grep -E "passed|failed|rror|real" gpurun_out/r05_gputests_full.log | tail -4 | cut -c1-300
python tools/soak_parity.py 220000 3000 2>&1 | tail -2 > gpurun_out/r05_soak_c.txt; python tools/soak_deep.py 600 24 2>&1 | tail -1 >> gpurun_out/r05_soak_c.txt; cat gpurun_out/r05_soak_c.txt
python3 tools/kernel_resources.py > gpurun_out/r05_kernel_resources.txt 2>&1
bash tools/dbg/bench_final.sh

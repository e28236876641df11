cd /root/repo
timeout 1800 python -m pytest tests/test_gpu_parity.py -x -q > gpurun_out/r05_gputests_d.log 2>&1; grep -E "passed|failed|rror" gpurun_out/r05_gputests_d.log | tail -3 | cut -c1-300
for i in 1 2; do
echo "== first bounce on constants"; python tools/bench_configs.py C4 2>&1 | tail -1
echo "== off"; SHM_LEAN_FIRST_BOUNCE=0 python tools/bench_configs.py C4 2>&1 | tail -1
done
bash tools/dbg/exp_leave.sh

cd /root/repo
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "render_parity or c4_frame or ab_switches" 2>&1 | tail -2
for i in 1 2; do
echo "== sorted tail"; python tools/bench_configs.py "C4" 2>&1 | tail -1
echo "== unsorted tail"; SHM_TAIL_SORT=0 python tools/bench_configs.py "C4" 2>&1 | tail -1
done

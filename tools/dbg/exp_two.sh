cd /root/repo
SHM_TRACE_TWO=1 timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "trace_bitwise or tree_shapes or edge_cases or render_parity" 2>&1 | tail -3
echo "== k_trace5"; python tools/bench_configs.py "S3 headline" 2>&1 | tail -1
echo "== k_trace6 (two rays per lane)"; SHM_TRACE_TWO=1 python tools/bench_configs.py "S3 headline" 2>&1 | tail -1
for rm in 32 48 64; do for lm in 16 24 32; do echo "== two: refill_min $rm leaf_min $lm"; SHM_TRACE_TWO=1 SHM_REFILL_MIN=$rm SHM_LEAF_MIN=$lm SHM_LEAF_MIN_ANY=$lm python tools/bench_configs.py "S3 headline" 2>&1 | tail -1; done; done

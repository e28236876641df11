cd /root/repo
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "split_their_plain" 2>&1 | grep -E "passed|failed" | tail -1
python tools/soak_parity.py 230000 4000 2>&1 | tail -2
python tools/soak_deep.py 700 32 2>&1 | tail -1
python tools/soak_trace.py 8 1048576 2>&1 | tail -1
for v in patch_emitter one_sphere instanced; do python tools/soak_trace.py 4 1048576 $v 2>&1 | tail -1; done

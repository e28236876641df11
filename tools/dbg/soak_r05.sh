cd /root/repo
python tools/soak_parity.py 200000 3000 2>&1 | tail -3
python tools/soak_deep.py 500 24 2>&1 | tail -1
python tools/soak_trace.py 8 1048576 2>&1 | tail -1
for v in patch_emitter one_sphere instanced; do python tools/soak_trace.py 4 1048576 $v 2>&1 | tail -1; done

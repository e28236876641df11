cd /root/repo
timeout 900 python3 -m pytest tests/test_gpu_parity.py -q -p no:cacheprovider -k "render_parity and (mesh_emitter or textured_object or smooth)" 2>&1 | grep -E "passed|failed"
python3 tools/film_ab.py --scenes S3to,S3t --rounds 2 "" 2>&1 | grep -v "^$"

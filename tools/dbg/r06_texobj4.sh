cd /root/repo
timeout 1200 python3 -m pytest tests/test_gpu_parity.py -q -p no:cacheprovider -k "textured or split or mesh_emitter or smooth" 2>&1 | grep -E "passed|failed"
python3 tools/film_ab.py --scenes S3to,S3t,C2t --rounds 2 "" 2>&1 | grep -v "^$"

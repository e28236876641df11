cd /root/repo
timeout 900 python3 -m pytest tests/test_gpu_parity.py -q -p no:cacheprovider -k "quads" 2>&1 | grep -E "passed|failed"
python3 tools/film_ab.py --scenes S3q --rounds 2 "" "SHM_OTHER_MIN=8" "SHM_OTHER_MIN=24" "SHM_OTHER_MIN=32" "SHM_OTHER_MIN=48" "SHM_OTHER_MIN=64" "SHM_REFILL_MIN=32" "SHM_REFILL_MIN=24" "SHM_LEAF_MIN=8,SHM_LEAF_MIN_ANY=4" 2>&1 | grep -v "^$"
python3 tools/film_ab.py --scenes S3,S3qc,S3c --rounds 1 "" 2>&1 | grep -v "^$"

cd /root/repo
timeout 900 python3 -m pytest tests/test_gpu_parity.py -q -p no:cacheprovider -k "both_occupancies" 2>&1 | grep -E "passed|failed"
python3 tools/film_ab.py --scenes S3q,S3q25 --rounds 1 "" "SHM_REFILL_MIN=32" "SHM_REFILL_MIN=24" "SHM_REFILL_MIN=32,SHM_LEAF_MIN=8" "SHM_REFILL_MIN=32,SHM_LEAF_MIN=8,SHM_LEAF_MIN_ANY=4" "SHM_REFILL_MIN=32,SHM_REFILL_MIN_ANY=24" "SHM_REFILL_MIN=36,SHM_REFILL_MIN_ANY=28" "SHM_REFILL_MIN=32,SHM_OTHER_MIN=12" "SHM_REFILL_MIN=32,SHM_OTHER_MIN=20" 2>&1 | grep -v "^$"

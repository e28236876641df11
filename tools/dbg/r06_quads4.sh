cd /root/repo
timeout 900 python3 -m pytest tests/test_gpu_parity.py -q -p no:cacheprovider -k "quads or patches or three_spheres or trace_tree" 2>&1 | grep -E "passed|failed"
python3 tools/film_ab.py --scenes S3q,S3q50,S3q25,S3q10,S3q03 --rounds 1 "SHM_GEN_HEAVY=0" "SHM_GEN_HEAVY=1" 2>&1 | grep -v "^$"
python3 tools/film_ab.py --scenes S3q --rounds 1 "SHM_OTHER_MIN=8" "SHM_OTHER_MIN=12" "SHM_OTHER_MIN=20" "SHM_LEAF_MIN=8" "SHM_REFILL_MIN=32" "SHM_REFILL_MIN=48" 2>&1 | grep -v "^$"

# the five-wave kernels' thresholds again now that a parked round's save / restore is in LDS (defaults: other 16, refill 24, leaf 16 / 8)
cd /root/repo
python3 tools/film_ab.py --scenes S3q --rounds 1 "" "SHM_OTHER_MIN=8" "SHM_OTHER_MIN=12" "SHM_OTHER_MIN=20" "SHM_REFILL_MIN=16" "SHM_REFILL_MIN=32" "SHM_LEAF_MIN=8" "SHM_LEAF_MIN=24" "SHM_OTHER_MIN=12,SHM_REFILL_MIN=32"
python3 tools/film_ab.py --scenes S3q25 --rounds 1 "" "SHM_OTHER_MIN=12" "SHM_OTHER_MIN=20" "SHM_REFILL_MIN=32"

cd /root/repo
python3 tools/film_ab.py --scenes S3q --rounds 2 "SHM_OTHER_MIN=28" "SHM_OTHER_MIN=36" "SHM_OTHER_MIN=40" "SHM_OTHER_MIN=44" "SHM_OTHER_MIN=40,SHM_OTHER_MIN_ANY=28" "SHM_OTHER_MIN=40,SHM_OTHER_MIN_ANY=32" "SHM_OTHER_MIN=40,SHM_OTHER_MIN_ANY=36" "SHM_OTHER_MIN=40,SHM_OTHER_MIN_ANY=32,SHM_REFILL_MIN=32" "SHM_OTHER_MIN=40,SHM_OTHER_MIN_ANY=32,SHM_REFILL_MIN=48" 2>&1 | grep -v "^$"
for C in 1 2; do SHM_LIB=shimmer_amd/csrc/_exp/lib_k_trace_census$C.so python3 tools/film_ab.py --scenes S3q --rounds 1 "" "SHM_OTHER_MIN=40,SHM_OTHER_MIN_ANY=32" 2>&1 | grep -v "^$"; done

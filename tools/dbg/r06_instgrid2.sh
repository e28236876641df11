# the instance-grid class: lane census of the closest-hit kernel (census build), then the parked-round / refill thresholds on the shipped library
cd /root/repo
SHM_LIB=shimmer_amd/csrc/_exp/lib_k_trace_census1.so python3 tools/film_ab.py --scenes S3ig,S3i --rounds 1 "" 2>&1 | grep -v "^\[rccl\|NCCL"
python3 tools/film_ab.py --scenes S3ig --rounds 1 "" "SHM_OTHER_MIN=8" "SHM_OTHER_MIN=24" "SHM_OTHER_MIN=32" "SHM_OTHER_MIN=40" "SHM_REFILL_MIN=24" "SHM_REFILL_MIN=8" "SHM_LEAF_MIN=8"

cd /root/repo
run() { L=$1; shift; echo "=== lib ${L:-product}"; if [ -n "$L" ]; then export SHM_LIB=shimmer_amd/csrc/_exp/lib_k_trace_$L.so; else unset SHM_LIB; fi; python tools/film_ab.py --scenes S3,C4 --rounds 2 "$@" 2>&1 | grep -v "^$"; }
run "" ""
run pl ""
run pf ""
run plf ""
run qp ""
run "" "" "SHM_REFILL_MIN=32" "SHM_REFILL_MIN=24"
run pb8 "" "SHM_REFILL_MIN=32" "SHM_REFILL_MIN=24" "SHM_REFILL_MIN=16"
run pb7 "" "SHM_REFILL_MIN=32" "SHM_REFILL_MIN=24" "SHM_REFILL_MIN=16"

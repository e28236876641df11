cd /root/repo
run() { L=$1; shift; echo "=== lib ${L:-product}"; if [ -n "$L" ]; then export SHM_LIB=shimmer_amd/csrc/_exp/lib_k_trace_$L.so; else unset SHM_LIB; fi; python3 tools/film_ab.py "$@" 2>&1 | grep -v "^$"; }
for L in "" g6 g5 g4; do
  run "$L" --scenes S3q --rounds 1 "" "SHM_OTHER_MIN=24" "SHM_OTHER_MIN=40,SHM_OTHER_MIN_ANY=32"
done
for L in "" g5 g4; do
  run "$L" --scenes S3p,S3s,S3i --rounds 1 ""
done

cd /root/repo
for L in "" nodefer7 nodefer8 defer7; do
  echo "=== lib ${L:-product(defer8)}"
  if [ -n "$L" ]; then export SHM_LIB=shimmer_amd/csrc/_exp/lib_k_trace_$L.so; else unset SHM_LIB; fi
  python tools/film_ab.py --scenes S3,C4 --rounds 2 "SHM_ANY_ORDER_FREE=0" "SHM_LEAF_MIN_FAST=8" "SHM_LEAF_MIN_FAST=12" 2>&1 | tail -12
done

# A/B: the five-wave general traversal kernels with their quadric / patch save area in LDS — 9 stack levels (shipped) against 10 (32 KB per workgroup, the whole LDS at five workgroups) —,
# and where the five-wave build now overtakes the seven-wave one (SHM_GEN_HEAVY override on partly-patch scenes); films must be bit-equal
cd /root/repo
for L in "" shimmer_amd/csrc/_exp/lib_k_trace_lds10.so; do
  echo "== library: ${L:-shipped}"
  SHM_LIB=$L python3 tools/film_ab.py --scenes S3q,S3qc --rounds 2 ""
done
python3 tools/film_ab.py --scenes S3q50,S3q25,S3q10,S3p --rounds 2 "SHM_GEN_HEAVY=0" "SHM_GEN_HEAVY=1"

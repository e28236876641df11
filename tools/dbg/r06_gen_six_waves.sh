# the seven-wave general traversal kernels (72 VGPRs + 189 / 79 spilled) at SIX waves (80 + 69 / 64) on the classes with FEW non-triangles and with instances
cd /root/repo
for L in "" shimmer_amd/csrc/_exp/lib_k_trace_c6.so shimmer_amd/csrc/_exp/lib_k_trace_c6a6.so; do
  echo "== library: ${L:-shipped}"
  SHM_LIB=$L python3 tools/film_ab.py --scenes S3i,S3ig,S3s,S3p --rounds 1 ""
done

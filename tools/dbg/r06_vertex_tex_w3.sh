# experiment: the textured vertex kernel at three waves per SIMD (168 VGPRs + 266 spilled, 1 056 B of scratch, 47.6 KB of LDS) against the shipped two-wave build (256 + 34, 576 B)
cd /root/repo
for L in "" shimmer_amd/csrc/_exp/lib_k_vertex_tex_w3.so; do
  echo "== library: ${L:-shipped}"
  SHM_LIB=$L python3 tools/film_ab.py --scenes C2t,S3t,S3to --rounds 2 ""
done

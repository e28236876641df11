# experiment: the textured vertex kernel with the next entry's hit word fetched ahead (K_VERTEX_PREFETCH) against the shipped build
cd /root/repo
for L in "" shimmer_amd/csrc/_exp/lib_k_vertex_tex_pf.so; do
  echo "== library: ${L:-shipped}"
  SHM_LIB=$L python3 tools/film_ab.py --scenes C2t,S3t,S3to --rounds 2 ""
done

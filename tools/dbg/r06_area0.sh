# experiment: the five-wave general kernels with the OUTER ray's save area (instance entry / leave) in LDS instead of the quadric / patch one, on the instance classes
cd /root/repo
SHM_LIB=shimmer_amd/csrc/_exp/lib_k_trace_area0.so python3 tools/film_ab.py --scenes S3i,S3ig --rounds 2 "SHM_GEN_HEAVY=0" "SHM_GEN_HEAVY=1"

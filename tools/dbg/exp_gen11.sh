cd /root/repo
SHM_LIB=shimmer_amd/csrc/_exp/lib_k_trace_K5_CENSUS_1.so python tools/bench_configs.py "S3i " 2>&1 | tail -5

cd /root/repo
timeout 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_pbrt_example.py -q -p no:cacheprovider -k "quads or patches or three_spheres or trace_tree or both_occupancies or mixed_shape or example_scene" 2>&1 | grep -E "passed|failed"
python3 tools/film_ab.py --scenes S3q,S3q25,S3p,S3s,S3qc --rounds 2 "" 2>&1 | grep -v "^$"
SHM_LIB=shimmer_amd/csrc/_exp/lib_k_trace_nosave.so python3 tools/film_ab.py --scenes S3q --rounds 2 "" 2>&1 | grep -v "^$"

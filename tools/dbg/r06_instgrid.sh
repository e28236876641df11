# class probe: the object as 4 x 4 x 4 instances of one small definition (ganesha_proxy(variant="instance_grid")): parity at small size, then the full-size frame at both
# occupancies of the general traversal kernels
cd /root/repo
python3 -m pytest tests/test_gpu_parity.py -q -m gpu -k "instance_grid" 2>&1 | tail -3
python3 tools/film_ab.py --scenes S3ig,S3i --rounds 2 "SHM_GEN_HEAVY=0" "SHM_GEN_HEAVY=1"

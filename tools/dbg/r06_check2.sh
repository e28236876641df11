# after the LDS save area (five-wave kernels), the 1/12 threshold and the instance scenes' parked-round threshold: the whole GPU suite, the bench line, node visits per ray of the instance classes
cd /root/repo
python3 -m pytest tests -q -m gpu -x 2>&1 | grep -E "passed|failed|error" | tail -3
python3 bench.py > gpurun_out/r06_bench_final2.json 2> gpurun_out/r06_bench_final2.err
python3 tools/film_ab.py --scenes S3,S3i,S3ig,S3q25 --rounds 1 ""

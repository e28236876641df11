cd /root/repo
export TMPDIR=/tmp
timeout 2400 python3 -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | grep -E "passed|failed|rror" | head -5
python3 bench.py > gpurun_out/r06_bench_final2.json 2> gpurun_out/r06_bench_final2.log
tail -20 gpurun_out/r06_bench_final2.log

cd /root/repo
export TMPDIR=/tmp
timeout 1800 python3 -m pytest tests -m gpu -q -x -p no:cacheprovider 2>&1 | tail -4
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python3 bench.py --no-live-pmc > gpurun_out/r06_bench_check.json 2> gpurun_out/r06_bench_check.log; tail -3 gpurun_out/r06_bench_check.log

cd /root/repo
export TMPDIR=/tmp
timeout 2400 python3 -m pytest tests -m gpu -q -x -p no:cacheprovider > gpurun_out/r06_gpu_suite.log 2>&1
tail -5 gpurun_out/r06_gpu_suite.log
bash tools/kernel_coverage.sh > gpurun_out/cov.log 2>&1
tail -15 gpurun_out/kernel_coverage.txt

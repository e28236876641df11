cd /root/repo
echo "== headline HIT16=0"; SHM_HIT16=0 python tools/bench_configs.py "S3 headline" 2>&1 | tail -1
echo "== headline GEN kernels (HIT16=0)"; SHM_TRACE_GEN=1 SHM_HIT16=0 python tools/bench_configs.py "S3 headline" 2>&1 | tail -1
python tools/bench_configs.py "S3s " "S3i " "S3p " 2>&1 | tail -3
for om in 8 32 48; do echo "== SHM_OTHER_MIN=$om"; SHM_OTHER_MIN=$om python tools/bench_configs.py "S3s " "S3i " "S3p " 2>&1 | tail -3; done

cd /root/repo
for V in 8 9 10 12 8 9; do echo "== SHM_LEAF_MIN_ANY=$V"; SHM_LEAF_MIN_ANY=$V python tools/bench_configs.py "S3 headline" 2>&1 | tail -1; done

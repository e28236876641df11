cd /root/repo
for R in 40 32 48; do for L in 16 12 24; do
echo "== refill_min $R leaf_min $L"; SHM_REFILL_MIN=$R SHM_LEAF_MIN=$L python tools/bench_configs.py S3s S3i 2>&1 | tail -2 | cut -c1-200
done; done

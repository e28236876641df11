cd /root/repo
for W in 2 4 8; do python3 tools/shard_balance.py --world $W 2>&1 | tail -1; done

cd /root/repo
python3 tools/soak_parity.py 300000 6000 2>&1 | tail -5
python3 tools/soak_deep.py 900 48 2>&1 | tail -2

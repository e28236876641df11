cd /root/repo
python tools/soak_parity.py 210000 3000 2>&1 | tail -3
python tools/soak_deep.py 500 24 2>&1 | tail -1

# at HEAD after the quirk switch grew (triangle and patch emitters as PBRT-v4 samples them): random scenes in BOTH settings, GPU against the oracle
cd /root/repo
python3 tools/soak_parity.py 309000 2000 1 2>&1 | tail -3
python3 tools/soak_parity.py 311000 1500 2>&1 | tail -3

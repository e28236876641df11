cd /root/repo
python3 tools/film_ab.py --scenes S3to --rounds 2 "" "SHM_SPLIT_PASS=1" 2>&1 | grep -v "^$"
python3 tools/film_ab.py --scenes C2t --rounds 2 "" "SHM_SPLIT_PASS=1" 2>&1 | grep -v "^$"

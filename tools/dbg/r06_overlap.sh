cd /root/repo
python3 tools/film_ab.py --scenes S3qc,S3c --rounds 2 "" "SHM_OVERLAP_PATHS=0" 2>&1 | grep -v "^$"

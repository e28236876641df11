# the parked-round threshold (default 16 lanes) at 24 / 32 across the scene classes with non-triangles
cd /root/repo
python3 tools/film_ab.py --scenes S3i,S3s,S3p,S3q,S3q10,S3ig --rounds 1 "" "SHM_OTHER_MIN=24" "SHM_OTHER_MIN=32"

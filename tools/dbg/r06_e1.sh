cd /root/repo
python tools/film_ab.py --scenes S3small,C4small,C2 --check-only "" "SHM_ANY_ORDER_FREE=0" "SHM_LEAF_MIN_FAST=8" "SHM_LEAF_MIN_FAST=48" 2>&1 | tail -20
python tools/film_ab.py --scenes S3,C4 --rounds 2 "SHM_ANY_ORDER_FREE=0" "" "SHM_LEAF_MIN_FAST=8" "SHM_LEAF_MIN_FAST=16" "SHM_LEAF_MIN_FAST=32" "SHM_LEAF_MIN_FAST=40" "SHM_LEAF_MIN_FAST=48" "SHM_LEAF_MIN_FAST=32,SHM_REFILL_MIN_FAST=32" "SHM_LEAF_MIN_FAST=32,SHM_REFILL_MIN_FAST=24" 2>&1 | tail -60

cd /root/repo
export TMPDIR=/tmp
timeout 900 python3 -m pytest tests/test_gpu_leaf_replay.py -x -q -p no:cacheprovider 2>&1 | tail -15

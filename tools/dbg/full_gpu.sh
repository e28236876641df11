cd /root/repo
( time timeout 3000 python -m pytest tests -x -q -m gpu ) > gpurun_out/r05_gputests_full.log 2>&1; grep -E "passed|failed|rror|real" gpurun_out/r05_gputests_full.log | tail -4 | cut -c1-300
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2

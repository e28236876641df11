cd /root/repo
PMC=1 tools/profile_side.sh r05 S3s S3i S3p > gpurun_out/r05_prof_side2.log 2>&1
python3 tools/kernel_resources.py > gpurun_out/r05_kernel_resources.txt 2>&1
python3 tools/c4_bounces.py > gpurun_out/r05_c4_per_bounce.txt 2>&1
tail -40 gpurun_out/r05_prof_side2.log
for W in 2 4 8; do python3 tools/shard_balance.py --world $W; done > gpurun_out/r05_shard_balance.txt 2>&1
tail -30 gpurun_out/r05_shard_balance.txt

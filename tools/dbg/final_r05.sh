cd /root/repo
PMC=1 tools/profile_side.sh r05 C4 > gpurun_out/r05_prof_side5.log 2>&1
python3 tools/kernel_resources.py > gpurun_out/r05_kernel_resources.txt 2>&1
python3 tools/c4_bounces.py > gpurun_out/r05_c4_per_bounce.txt 2>&1
bash tools/dbg/bench_final.sh

cd /root/repo
PMC=1 bash tools/profile_side.sh r06 S3to > gpurun_out/r06_texobj2_side.log 2>&1
cat gpurun_out/prof_r06_S3to/summary.txt | head -120

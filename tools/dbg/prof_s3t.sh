cd /root/repo
PMC=1 tools/profile_side.sh r05 S3t > gpurun_out/r05_prof_s3t.log 2>&1
grep -E "calls=" gpurun_out/prof_r05_S3t/summary.txt | head -16
grep -E "lanes per VALU" gpurun_out/prof_r05_S3t/summary.txt | head -16

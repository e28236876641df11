cd /root/repo
tools/profile_gpu.sh r05_end --spp 256 --steps 1 --warmup 0 --no-cpu-baseline --no-side --no-live-pmc > gpurun_out/r05_prof_end.log 2>&1
tools/profile_side.sh r05 "S3s " "S3i " "S3p " "S3c" "C4" "C2t" > gpurun_out/r05_prof_side.log 2>&1
tail -5 gpurun_out/r05_prof_end.log; tail -30 gpurun_out/r05_prof_side.log

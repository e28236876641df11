# the round's closing measurements on one box: the bench line as the driver runs it (live counters, CPU baseline, side results), then the rocprofv3 passes of the same frame
cd /root/repo
export TMPDIR=/tmp
python3 bench.py > gpurun_out/r06_bench_final.json 2> gpurun_out/r06_bench_final.log
tail -3 gpurun_out/r06_bench_final.log
bash tools/profile_gpu.sh r06_end --spp 256 --steps 1 --warmup 0 --no-cpu-baseline --no-side --no-live-pmc > gpurun_out/r06_end_profile.log 2>&1
cp gpurun_out/prof_r06_end/summary.txt gpurun_out/r06_end_spp256.txt
head -12 gpurun_out/r06_end_spp256.txt

cd /root/repo
( time python bench.py ) > gpurun_out/r05_bench_final.log 2>&1
grep '^{' gpurun_out/r05_bench_final.log | tail -1 > gpurun_out/r05_bench_final.json
( time python bench.py --force-dist --width 3840 --height 2160 --spp 1024 --steps 1 --warmup 1 --no-cpu-baseline --no-side ) > gpurun_out/r05_bench_c5_world1.log 2>&1
grep '^{' gpurun_out/r05_bench_c5_world1.log | tail -1 > gpurun_out/r05_bench_c5_world1.json
tail -3 gpurun_out/r05_bench_final.log | cut -c1-1500; tail -3 gpurun_out/r05_bench_c5_world1.log | cut -c1-600

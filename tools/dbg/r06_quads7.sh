cd /root/repo
for C in 1 2; do SHM_LIB=shimmer_amd/csrc/_exp/lib_k_trace_census$C.so python3 tools/film_ab.py --scenes S3q --rounds 1 "" 2>&1 | grep -v "^$"; done
export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d gpurun_out/prof_r06_q/pmc_a -o pmc -- python3 tools/bench_configs.py S3q > gpurun_out/prof_r06_q_a.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d gpurun_out/prof_r06_q/pmc_FETCH_SIZE -o pmc -- python3 tools/bench_configs.py S3q > gpurun_out/prof_r06_q_b.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d gpurun_out/prof_r06_q/pmc_WRITE_SIZE -o pmc -- python3 tools/bench_configs.py S3q > gpurun_out/prof_r06_q_c.log 2>&1
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r06_q/stats -o stats -- python3 tools/bench_configs.py S3q > gpurun_out/prof_r06_q/bench.log 2>&1
python3 tools/summarize_prof.py gpurun_out/prof_r06_q > gpurun_out/prof_r06_q_summary.txt 2>&1
grep -E "k_trace5" gpurun_out/prof_r06_q_summary.txt | head -40

#!/bin/bash
# Profile the bench workload on the GPU box (run through gpurun). Kernel-trace/stats and each PMC set are separate
# passes (MI355X_MICROARCH.md: never combine --pmc with trace domains; TCC has 4 slots, FETCH_SIZE uses 3).
# Usage: tools/profile_gpu.sh <tag> [bench args...]   -> gpurun_out/prof_<tag>/
set -u
TAG=${1:-r01}; shift || true
ARGS=${@:---spp 16 --steps 1 --warmup 0 --no-cpu-baseline}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$(pwd)
cd $ROOT
rocprofv3 --kernel-trace --stats -d $OUT/stats -o stats -- python3 bench.py $ARGS > $OUT/bench_stats.json 2> $OUT/bench_stats.log
for SET in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INSTS_BRANCH" "GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM SQ_WAVES"; do
  NAME=$(echo $SET | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $SET -d $OUT/pmc_$NAME -o pmc -- python3 bench.py $ARGS > $OUT/bench_$NAME.json 2> $OUT/bench_$NAME.log
done
python3 tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt

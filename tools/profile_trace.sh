#!/bin/bash
# Issue / lane / wait counters of the traversal kernels. Usage: tools/profile_trace.sh <tag> [env assignments...] -- [bench args]
set -u
TAG=$1; shift
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ARGS="--spp 64 --steps 1 --warmup 0 --no-cpu-baseline --no-side"
rocprofv3 --kernel-trace --stats -d $OUT/stats -o stats -- python3 bench.py $ARGS > $OUT/bench_stats.json 2> $OUT/bench_stats.log
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS" \
           "SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INST_CYCLES_VMEM_RD" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM SQ_WAVES SQ_INST_LEVEL_LDS SQ_WAIT_INST_LDS"; do
  NAME=$(echo $SET | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $SET -d $OUT/pmc_$NAME -o pmc -- python3 bench.py $ARGS > $OUT/bench_$NAME.json 2> $OUT/bench_$NAME.log || echo "FAILED: $SET"
done
python3 tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
grep -E "k_trace" $OUT/summary.txt

#!/bin/bash
# Quick SQ-level PMC sets for whichever trace kernel the env selects. Usage: [env] tools/profile_pmc3.sh <tag>
TAG=${1:-x}; OUT=gpurun_out/prof_$TAG; mkdir -p $OUT; export TMPDIR=/tmp
ARGS="--spp 16 --steps 1 --warmup 0 --no-cpu-baseline"
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS" \
           "SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INST_CYCLES_VMEM_RD SQ_WAVES SQ_IFETCH" \
           "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TA_TA_BUSY_sum TA_BUSY_avr" "GRBM_GUI_ACTIVE"; do
  NAME=$(echo $SET | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $SET -d $OUT/pmc_$NAME -o pmc -- python3 bench.py $ARGS > $OUT/bench_$NAME.json 2> $OUT/bench_$NAME.log || echo "FAILED: $SET"
done
python3 tools/summarize_prof.py $OUT 2>&1 | grep -E "k_trace" | grep -v "^   "

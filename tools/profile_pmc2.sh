#!/bin/bash
# Second-level PMC sets for the traversal kernel (issue / TA / TCP / UTCL1 behaviour). Usage: tools/profile_pmc2.sh <tag> [bench args]
set -u
TAG=${1:-r01}; shift || true
ARGS=${@:---spp 16 --steps 1 --warmup 0 --no-cpu-baseline}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
for SET in "SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INST_CYCLES_VMEM_RD" \
           "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_BUSY_CU_CYCLES SQ_CYCLES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_LEVEL_WAVES SQ_IFETCH" \
           "TA_TA_BUSY_sum TA_BUSY_avr" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" "TA_FLAT_READ_WAVEFRONTS_sum TA_TOTAL_WAVEFRONTS_sum" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" "TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" \
           "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum" "TCP_UTCL1_REQUEST_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum" "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum" "TD_TD_BUSY_sum TD_TC_STALL_sum"; do
  NAME=$(echo $SET | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $SET -d $OUT/pmc_$NAME -o pmc -- python3 bench.py $ARGS > $OUT/bench_$NAME.json 2> $OUT/bench_$NAME.log || echo "FAILED: $SET"
done
python3 tools/summarize_prof.py $OUT 2>&1 | grep -E "k_trace|k_shade" > $OUT/summary2.txt
cat $OUT/summary2.txt

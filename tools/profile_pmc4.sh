#!/bin/bash
# VALU lane utilisation and instruction mix per kernel. Usage: tools/profile_pmc4.sh <tag> [bench args]
TAG=${1:-x}; shift || true
ARGS=${@:---spp 64 --steps 1 --warmup 0 --no-cpu-baseline}
OUT=gpurun_out/prof_$TAG; mkdir -p $OUT; export TMPDIR=/tmp
for SET in "SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" \
           "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_CVT"; do
  NAME=$(echo $SET | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $SET -d $OUT/pmc_$NAME -o pmc -- python3 bench.py $ARGS > $OUT/bench_$NAME.json 2> $OUT/bench_$NAME.log || echo "FAILED: $SET"
done
python3 tools/summarize_prof.py $OUT 2>&1 | grep -E "^k_shade|^k_trace3|^k_generate" | grep -v "^   "

#!/bin/bash
# A/B of two environment configurations under the SQ counter sets (separate passes): tools/dbg/pmc_ab.sh <tag> "<env A>" "<env B>" [bench args]
set -u
TAG=$1; A=$2; B=$3; shift 3
ARGS=${@:---spp 64 --steps 1 --warmup 0 --no-cpu-baseline --no-side --no-live-pmc}
export TMPDIR=/tmp
for CFG in A B; do
  ENVS=$A; [ $CFG = B ] && ENVS=$B
  OUT=gpurun_out/pmcab_${TAG}_$CFG
  mkdir -p $OUT
  for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INSTS_BRANCH" "SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_INSTS_SMEM SQ_WAVES SQ_INST_LEVEL_VMEM" "GRBM_GUI_ACTIVE"; do
    NAME=$(echo $SET | tr ' ' '_' | cut -c1-40)
    ( export $ENVS; rocprofv3 --pmc $SET -d $OUT/pmc_$NAME -o pmc -- python3 bench.py $ARGS > $OUT/bench_$NAME.json 2> $OUT/bench_$NAME.log )
  done
  echo "=== $CFG: $ENVS"; python3 tools/summarize_prof.py $OUT 2>&1 | grep -E "k_trace|==" 
done

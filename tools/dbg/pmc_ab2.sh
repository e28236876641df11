#!/bin/bash
# like pmc_ab.sh, with per-configuration bench arguments: tools/dbg/pmc_ab2.sh <tag> "<env A>|<args A>" "<env B>|<args B>"
set -u
TAG=$1; shift
export TMPDIR=/tmp
COMMON="--spp 64 --steps 1 --warmup 0 --no-cpu-baseline --no-side --no-live-pmc"
K=0
for SPEC in "$@"; do
  K=$((K+1))
  ENVS=${SPEC%%|*}; ARGS=${SPEC#*|}
  OUT=gpurun_out/pmcab_${TAG}_$K
  mkdir -p $OUT
  for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INSTS_BRANCH" "SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_INSTS_SMEM SQ_WAVES SQ_INST_LEVEL_VMEM"; do
    NAME=$(echo $SET | tr ' ' '_' | cut -c1-40)
    ( [ -n "$ENVS" ] && export $ENVS; rocprofv3 --pmc $SET -d $OUT/pmc_$NAME -o pmc -- python3 bench.py $COMMON $ARGS > $OUT/bench_$NAME.json 2> $OUT/bench_$NAME.log )
  done
  echo "=== $K: env [$ENVS] args [$ARGS]"; python3 tools/summarize_prof.py $OUT 2>&1 | grep -E "k_trace" 
done

#!/bin/bash
# Per-kernel times (rocprofv3 --kernel-trace --stats) of the side configurations of tools/bench_configs.py.
# Usage: [PMC=1] tools/profile_side.sh <tag> <config substring>...   -> gpurun_out/prof_<tag>_<config>/
set -u
TAG=$1; shift
export TMPDIR=/tmp
for CFG in "$@"; do
  OUT=gpurun_out/prof_${TAG}_${CFG}
  mkdir -p $OUT
  rocprofv3 --kernel-trace --stats -d $OUT/stats -o stats -- python3 tools/bench_configs.py $CFG > $OUT/bench.log 2>&1
  if [ "${PMC:-0}" = "1" ]; then  # issue / wait / lane counters (separate passes, never with a trace domain)
    for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INSTS_BRANCH" "FETCH_SIZE" "WRITE_SIZE"; do
      NAME=$(echo $SET | tr ' ' '_' | cut -c1-40)
      rocprofv3 --pmc $SET -d $OUT/pmc_$NAME -o pmc -- python3 tools/bench_configs.py $CFG > $OUT/bench_$NAME.log 2>&1
    done
  fi
  python3 tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
  echo "=== $CFG"; grep -v amdgpu.ids $OUT/bench.log | tail -1; grep -E "calls=" $OUT/summary.txt | head -14
done

#!/bin/bash
# Per-kernel times (rocprofv3 --kernel-trace --stats) of the side configurations of tools/bench_configs.py.
# Usage: tools/profile_side.sh <tag> <config substring>...   -> gpurun_out/prof_<tag>_<config>/
set -u
TAG=$1; shift
export TMPDIR=/tmp
for CFG in "$@"; do
  OUT=gpurun_out/prof_${TAG}_${CFG}
  mkdir -p $OUT
  rocprofv3 --kernel-trace --stats -d $OUT/stats -o stats -- python3 tools/bench_configs.py $CFG > $OUT/bench.log 2>&1
  python3 tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
  echo "=== $CFG"; grep -v amdgpu.ids $OUT/bench.log | tail -1; grep -E "calls=" $OUT/summary.txt | head -14
done

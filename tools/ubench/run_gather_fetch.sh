#!/bin/bash
# Runs tools/ubench/gather_fetch under rocprofv3 (FETCH_SIZE, then the TCC request counters, separate passes) on the GPU box and
# prints bytes reported per byte requested for every (pattern, array size). Usage (through gpurun): bash tools/ubench/run_gather_fetch.sh
set -u
export TMPDIR=/tmp
OUT=${1:-gpurun_out/gather_fetch}
mkdir -p $OUT
hipcc --offload-arch=gfx950 -O2 tools/ubench/gather_fetch.hip -o $OUT/gather_fetch || exit 1
timeout 300 $OUT/gather_fetch > $OUT/plain.txt || exit 1
for SET in "FETCH_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_READ_sum"; do
  NAME=$(echo $SET | tr ' ' '_')
  timeout 600 rocprofv3 --pmc $SET --output-format csv -d $OUT/pmc_$NAME -o pmc -- $OUT/gather_fetch > $OUT/run_$NAME.txt 2> $OUT/run_$NAME.log || echo "pass $SET failed (see $OUT/run_$NAME.log)"
done
python3 tools/ubench/summarize_gather_fetch.py $OUT | tee $OUT/summary.txt

#!/bin/bash
# Sweep the traversal kernel's thresholds on the headline frame. Usage: tools/sweep_trace.sh "<refill_min...>" "<leaf_min...>" "<pf_min...>" [bench args]
R=${1:-"4 8 16"}; L=${2:-"16"}; P=${3:-"32"}; shift 3 || true
for r in $R; do for l in $L; do for p in $P; do
  echo -n "refill_min=$r leaf_min=$l pf_min=$p : "
  SHM_REFILL_MIN=$r SHM_LEAF_MIN=$l SHM_PF_MIN=$p python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-side "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); b=d['breakdown_ms_per_step']; print('%.0f Mray/s  closest %.1f any %.1f shade %.1f' % (d['value'], b['trace_closest'], b['trace_any'], b['shade_generate_film']))"
done; done; done

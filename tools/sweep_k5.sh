#!/bin/bash
# A/B sweep of the pair-step traversal kernel on the headline frame (run through gpurun): every line is one bench.py run of 3 frames.
#   tools/sweep_k5.sh "<label>|<env assignments>" ...
one() {
  label=$1; shift
  env "$@" python bench.py --no-cpu-baseline --no-side --no-live-pmc --steps 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); b=d['breakdown_ms_per_step']
print('%-44s %7.1f Mray/s | closest %6.1f any %6.1f shade %6.1f total %6.1f ms' % ('$label', d['value'], b['trace_closest'], b['trace_any'], b['shade_generate_film'], b['gpu_total']))"
}
for spec in "$@"; do
  label=${spec%%|*}; envs=${spec#*|}
  one "$label" $envs
done

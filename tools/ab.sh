#!/bin/bash
# A/B helper: tools/ab.sh "<bench args>" "<env cfg 1>" "<env cfg 2>" ...
ARGS=$1; shift
for cfg in "$@"; do echo "== $cfg"; env $cfg python bench.py $ARGS --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=dict(d['roofline_hbm_algorithmic'], **{k: d['roofline'][k] for k in ('avg_launch_ms', 'launches')}); b=d['breakdown_ms_per_step']
print('value %.1f Mray/s | K2 %.0f GB/s (%.1f%%) %.0f Mray/s avg %.2f ms x%d | closest %.1f any %.1f shade %.1f total %.1f ms'%(d['value'],r['achieved'],100*r['frac'],r['closest_Mray_s_in_kernel'],r['avg_launch_ms'],r['launches'],b['trace_closest'],b['trace_any'],b['shade_generate_film'],b['gpu_total']))"; done

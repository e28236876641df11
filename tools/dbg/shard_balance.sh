#!/bin/bash
# Load balance of the tile sharding: each rank's share of the headline frame rendered alone on one GPU (max / mean = the scaling loss)
N=${1:-8}
for r in $(seq 0 $((N-1))); do
  python bench.py --shard-of $N --shard-rank $r --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('rank $r of $N: %.2f ms/step, %.0f Mray/s, rays %d' % (d['ms_per_step'], d['value'], d['config']['rays_per_step']))"
done

#!/bin/bash
# A/B of SHM_TRACE_RAYS_PER_LANE (how much of the persistent traversal grid a small queue uses) on the configurations with small launches
for R in 0 4 8 16 32; do
  echo "== SHM_TRACE_RAYS_PER_LANE=$R"
  SHM_TRACE_RAYS_PER_LANE=$R python3 tools/bench_configs.py "C2 cornell" "C4 crown" "C2t cornell" "E3 spheres" 2>&1 | grep -v "^$"
done

cd /root/repo
hipcc --offload-arch=gfx950 -O2 tools/ubench/valu_exec.hip -o tools/ubench/valu_exec 2>/dev/null
timeout 120 tools/ubench/valu_exec > gpurun_out/r05_valu_exec.txt 2>&1
cat gpurun_out/r05_valu_exec.txt
for B in 8 6 4 3 2 1; do echo "== SHM_TAIL_FUSED_BOUNCE=$B"; SHM_TAIL_FUSED_BOUNCE=$B python tools/bench_configs.py C4 2>&1 | tail -1; done

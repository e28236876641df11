cd /root/repo
hipcc --offload-arch=gfx950 -O2 tools/ubench/valu_exec.hip -o tools/ubench/valu_exec 2>/dev/null
timeout 120 tools/ubench/valu_exec > gpurun_out/r05_valu_exec.txt 2>&1
tail -36 gpurun_out/r05_valu_exec.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "fused_all_materials or ab_switches or c4_frame or render_parity" 2>&1 | tail -3
for i in 1 2; do for B in 1 0; do echo "== SHM_TAIL_FUSED_BOUNCE=$B"; SHM_TAIL_FUSED_BOUNCE=$B python tools/bench_configs.py C4 2>&1 | tail -1; done; done
for B in -1 1 0; do echo "== SHM_TAIL_FUSED_BOUNCE=$B"; SHM_TAIL_FUSED_BOUNCE=$B python tools/bench_configs.py C2p C1 2>&1 | tail -2; done

cd /root/repo
hipcc --offload-arch=gfx950 -O2 tools/ubench/valu_exec.hip -o tools/ubench/valu_exec 2>/dev/null
timeout 120 tools/ubench/valu_exec > gpurun_out/r05_valu_exec.txt 2>&1
grep -E "VOP" gpurun_out/r05_valu_exec.txt

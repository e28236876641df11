cd /root/repo
hipcc --offload-arch=gfx950 -O2 tools/ubench/valu_exec.hip -o tools/ubench/valu_exec 2>/dev/null
timeout 120 tools/ubench/valu_exec > gpurun_out/r05_valu_exec.txt 2>&1
tail -6 gpurun_out/r05_valu_exec.txt
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_multi.py -x -q 2>&1 | tail -3
python tools/bench_configs.py C4 C2 "S3 headline" S3c 2>&1 | grep -v amdgpu

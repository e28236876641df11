# at HEAD (five-wave kernels' LDS save area, 1/12 threshold, 32 parked lanes in instance scenes, threaded BVH build): another soak, then counter profiles of the patch object and the instance grid
cd /root/repo
python3 tools/soak_parity.py 306000 3000 2>&1 | tail -3
python3 tools/soak_deep.py 948 24 2>&1 | tail -2
PMC=1 bash tools/profile_side.sh r06 S3q S3ig
cp gpurun_out/prof_r06_S3q/summary.txt gpurun_out/r06_staged_S3q.txt
cp gpurun_out/prof_r06_S3ig/summary.txt gpurun_out/r06_staged_S3ig.txt

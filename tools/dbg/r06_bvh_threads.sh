# shm_bvh_build on the headline object's 4.3 M triangles: serial against host threads (the GPU box's host cores); node / order hashes must agree
cd /root/repo
nproc
for T in 1 4 16 64 0; do
  echo "== SHM_BVH_THREADS=$T (0: unset, the default)"
  if [ $T = 0 ]; then python3 tools/dbg/r06_bvh_threads.py 2>&1 | grep -v cube; else SHM_BVH_THREADS=$T python3 tools/dbg/r06_bvh_threads.py 2>&1 | grep -v cube; fi
done

#!/bin/bash
# Timing experiments: builds variants of ONE kernel translation unit with -D<macro>=<value> and links each into its own library under
# shimmer_amd/csrc/_exp/ (git-ignored), to be loaded with SHM_LIB=... by bench.py. The product library is not touched.
#   tools/exp_variants.sh k_scatter_layered_tri SHM_EXP_SKIP 1 2 4 8
set -e
cd "$(dirname "$0")/../shimmer_amd/csrc"
TU=$1; MACRO=$2; shift 2
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -fno-gpu-flush-denormals-to-zero -Wno-unused-result -Wno-unused-value"
mkdir -p _exp
for V in "$@"; do
  ( /opt/rocm/bin/hipcc $FLAGS -D$MACRO=$V -x hip -c $TU.hip -o _exp/${TU}_$V.o &&
    OBJS=$(ls _obj/*.o | grep -v "_obj/$TU.o") &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o _exp/lib_${TU}_${MACRO}_$V.so $OBJS _exp/${TU}_$V.o -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib ) &
done
wait
ls -la _exp/*.so

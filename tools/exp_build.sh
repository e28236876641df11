#!/bin/bash
# Timing experiments: variants of ONE kernel translation unit, each with its own -D flags, each linked into its own library under shimmer_amd/csrc/_exp/
# (git-ignored; the libraries travel to the GPU box) to be loaded with SHM_LIB=... The product library is not touched (but must be built: its other objects are reused).
#   tools/exp_build.sh k_trace nodefer7:"-DK5_FAST_DEFER=0 -DK5_FAST_WAVES=7" defer7:"-DK5_FAST_WAVES=7"      ->  _exp/lib_k_trace_nodefer7.so, _exp/lib_k_trace_defer7.so
set -e
cd "$(dirname "$0")/../shimmer_amd/csrc"
TU=$1; shift
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -fno-gpu-flush-denormals-to-zero -Wno-unused-result -Wno-unused-value"
mkdir -p _exp
for SPEC in "$@"; do
  NAME=${SPEC%%:*}; DEFS=${SPEC#*:}
  ( /opt/rocm/bin/hipcc $FLAGS $DEFS -x hip -c $TU.hip -o _exp/${TU}_$NAME.o &&
    OBJS=$(ls _obj/*.o | grep -v "_obj/$TU.o") &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o _exp/lib_${TU}_$NAME.so $OBJS _exp/${TU}_$NAME.o -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib ) &
done
wait
ls -la _exp/lib_${TU}_*.so

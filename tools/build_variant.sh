#!/bin/bash
# tools/build_variant.sh NAME [-DMACRO=VALUE ...]: compile shimmer_amd/csrc/variants/libshimmer_hip_NAME.so with extra
# defines for A/B runs on the GPU box (select with SHM_LIB=shimmer_amd/csrc/variants/libshimmer_hip_NAME.so).
set -e
cd "$(dirname "$0")/.."
NAME=$1; shift
mkdir -p shimmer_amd/csrc/variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fvisibility=hidden -ffp-contract=off -fno-fast-math \
  -fhip-fp32-correctly-rounded-divide-sqrt -fno-gpu-flush-denormals-to-zero -Wno-unused-result -Wno-unused-value "$@" \
  -x hip shimmer_amd/csrc/shimmer_hip.hip shimmer_amd/csrc/host/host_mirror.cpp -o shimmer_amd/csrc/variants/libshimmer_hip_$NAME.so
echo built shimmer_amd/csrc/variants/libshimmer_hip_$NAME.so

// gather_fetch.hip — what rocprofv3's FETCH_SIZE reports for the traversal kernels' access pattern on gfx950.
// MI355X_MICROARCH.md calibrates FETCH_SIZE (x2) for wide coalesced streams only and calls other widths uncalibrated. This program issues a
// KNOWN number of record fetches in three patterns over arrays below and far above the 256 MiB Infinity Cache:
//   stream    every lane reads 16 B, consecutive lanes consecutive addresses (the guide's calibration case)
//   gather32  every lane reads the two 16-B halves of one 32-B record at a pseudo-random index (k_trace3: one BVH node per step)
//   gather64  every lane reads the four 16-B quarters of one 64-B aligned block (the pair-step kernel: both children of a node)
//   gather128 every lane reads all eight 16-B pieces of one 128-B aligned line
//   split128  every lane reads bytes 0-15 and 64-79 of one 128-B aligned line: ONE request per line means the L2 fetches whole 128-B lines
//             on a miss, TWO mean it fetches 64-B halves — which decides what a gather request really moves (the counter tallies 64 B either way)
// Indices are a bijection of [0, n_records) (xorshift-multiply rounds on log2(n) bits), so one pass touches every record exactly once:
// requested bytes = array bytes, unique lines = the whole array at any granularity, and nothing is re-read within a pass.
// Run under `rocprofv3 --pmc FETCH_SIZE` (tools/ubench/run_gather_fetch.sh); each (pattern, size) is one dispatch, printed in launch order.
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/gather_fetch.hip -o tools/ubench/gather_fetch
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ uint32_t permute_bits(uint32_t i, int bits) {  // a bijection of [0, 2^bits)
    const uint32_t mask = (bits == 32) ? 0xffffffffu : ((1u << bits) - 1u);
    uint32_t x = i & mask;
    x = (x * 0x9E3779B1u) & mask;   // odd multiplier: invertible mod 2^bits
    x ^= x >> (bits / 2 + 1);       // xorshift by more than half the width: invertible
    x = (x * 0x85EBCA6Bu) & mask;
    x ^= x >> (bits / 2 + 2);
    x = (x * 0xC2B2AE35u) & mask;
    x ^= x >> (bits / 2 + 1);
    return x & mask;
}

__global__ void __launch_bounds__(256) k_stream(const float4* __restrict__ a, size_t n16, float* out) {
    float acc = 0.0f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) {
        const float4 v = a[i];
        acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 12345.678f) out[0] = acc;
}
__global__ void __launch_bounds__(256) k_gather32(const float4* __restrict__ a, uint32_t n_rec, int bits, float* out) {
    float acc = 0.0f;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_rec; i += gridDim.x * blockDim.x) {
        const float4* r = a + 2u * (size_t)permute_bits(i, bits);
        const float4 v0 = r[0], v1 = r[1];
        acc += v0.x + v0.w + v1.y + v1.z;
    }
    if (acc == 12345.678f) out[0] = acc;
}
__global__ void __launch_bounds__(256) k_gather64(const float4* __restrict__ a, uint32_t n_blk, int bits, float* out) {
    float acc = 0.0f;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_blk; i += gridDim.x * blockDim.x) {
        const float4* r = a + 4u * (size_t)permute_bits(i, bits);
        const float4 v0 = r[0], v1 = r[1], v2 = r[2], v3 = r[3];
        acc += v0.x + v1.w + v2.y + v3.z;
    }
    if (acc == 12345.678f) out[0] = acc;
}

__global__ void __launch_bounds__(256) k_gather128(const float4* __restrict__ a, uint32_t n_blk, int bits, float* out) {
    float acc = 0.0f;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_blk; i += gridDim.x * blockDim.x) {
        const float4* r = a + 8u * (size_t)permute_bits(i, bits);
        const float4 v0 = r[0], v1 = r[1], v2 = r[2], v3 = r[3], v4 = r[4], v5 = r[5], v6 = r[6], v7 = r[7];
        acc += v0.x + v1.w + v2.y + v3.z + v4.x + v5.w + v6.y + v7.z;
    }
    if (acc == 12345.678f) out[0] = acc;
}
__global__ void __launch_bounds__(256) k_split128(const float4* __restrict__ a, uint32_t n_blk, int bits, float* out) {
    float acc = 0.0f;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_blk; i += gridDim.x * blockDim.x) {
        const float4* r = a + 8u * (size_t)permute_bits(i, bits);
        const float4 v0 = r[0], v4 = r[4];
        acc += v0.x + v4.w;
    }
    if (acc == 12345.678f) out[0] = acc;
}

int main() {
    float* d_out;
    CHECK(hipMalloc(&d_out, 64));
    const int sizes_mb[] = {64, 512, 4096};
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    int dispatch = 0;
    for (int si = 0; si < 3; ++si) {
        const size_t bytes = (size_t)sizes_mb[si] << 20;
        float4* d;
        CHECK(hipMalloc(&d, bytes));
        CHECK(hipMemset(d, 0, bytes));
        CHECK(hipDeviceSynchronize());
        int bits32 = 0, bits64 = 0, bits128 = 0;
        while (((size_t)32 << bits32) < bytes) ++bits32;
        while (((size_t)64 << bits64) < bytes) ++bits64;
        while (((size_t)128 << bits128) < bytes) ++bits128;
        for (int pat = 0; pat < 5; ++pat) {
            for (int rep = 0; rep < 2; ++rep) {  // the first pass of a pattern also warms the caches with whatever fits; both are reported
                CHECK(hipEventRecord(e0));
                if (pat == 0) hipLaunchKernelGGL(k_stream, dim3(256 * 16), dim3(256), 0, 0, d, bytes / 16, d_out);
                else if (pat == 1) hipLaunchKernelGGL(k_gather32, dim3(256 * 16), dim3(256), 0, 0, d, (uint32_t)(bytes / 32), bits32, d_out);
                else if (pat == 2) hipLaunchKernelGGL(k_gather64, dim3(256 * 16), dim3(256), 0, 0, d, (uint32_t)(bytes / 64), bits64, d_out);
                else if (pat == 3) hipLaunchKernelGGL(k_gather128, dim3(256 * 16), dim3(256), 0, 0, d, (uint32_t)(bytes / 128), bits128, d_out);
                else hipLaunchKernelGGL(k_split128, dim3(256 * 16), dim3(256), 0, 0, d, (uint32_t)(bytes / 128), bits128, d_out);
                CHECK(hipEventRecord(e1));
                CHECK(hipEventSynchronize(e1));
                float ms = 0;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                static const char* names[5] = {"stream", "gather32", "gather64", "gather128", "split128"};
                const char* name = names[pat];
                const size_t rec_bytes[5] = {16, 32, 64, 128, 128};
                const size_t req = pat == 4 ? bytes / 4 : bytes;  // split128 asks for 32 of each line's 128 bytes
                printf("dispatch %d pattern %s array_MB %d pass %d requested_bytes %zu records %zu ms %.4f requested_GBs %.1f\n", dispatch++, name, sizes_mb[si], rep, req,
                       bytes / rec_bytes[pat], ms, req / (ms * 1e-3) / 1e9);
            }
        }
        CHECK(hipFree(d));
    }
    return 0;
}

// Gather-rate microbenchmark for gfx950: how many 16-B lane-loads per second can the chip sustain when every lane
// reads a different random record?  Variants model the BVH access patterns considered in DESIGN.md.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ unsigned lcg(unsigned& s) { s = s * 1664525u + 1013904223u; return s; }

// mode 0: lane loads 2x16B of its own random 32-B record (today's node fetch)
// mode 1: lane loads 1x16B of its own random 32-B record
// mode 2: lane PAIR shares a random 32-B record, each lane loads one 16-B half (one instruction, 32 distinct records)
// mode 3: lane QUAD shares a random 64-B block, each lane loads one 16-B quarter (one instruction, 16 distinct blocks)
// mode 4: lane loads 4x16B = one whole random 64-B block (sibling pair fetched by one lane)
template <int MODE>
__global__ void __launch_bounds__(256) k_gather(const float4* __restrict__ data, unsigned n_rec32, int iters, float* out) {
    unsigned lane = threadIdx.x & 63;
    unsigned tid = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned s;
    if (MODE == 2) s = (tid >> 1) * 747796405u + 12345u;
    else if (MODE == 3) s = (tid >> 2) * 747796405u + 12345u;
    else s = tid * 747796405u + 12345u;
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {  // 4 independent records in flight per lane
            unsigned r = lcg(s) % n_rec32;
            if (MODE == 0) { float4 a = data[2ull * r], b = data[2ull * r + 1]; acc += a.x + b.w; }
            if (MODE == 1) { float4 a = data[2ull * r]; acc += a.x; }
            if (MODE == 2) { float4 a = data[2ull * r + (lane & 1)]; acc += a.x; }
            if (MODE == 3) { unsigned r64 = r >> 1; float4 a = data[4ull * r64 + (lane & 3)]; acc += a.x; }
            if (MODE == 4) { unsigned r64 = r >> 1; float4 a = data[4ull * r64], b = data[4ull * r64 + 1], c = data[4ull * r64 + 2], d = data[4ull * r64 + 3]; acc += a.x + b.y + c.z + d.w; }
        }
    }
    if (acc == 123.456f) out[0] = acc;
}

template <int MODE>
void run(const char* name, const float4* d, size_t bytes, int blocks, int iters, float* dout, int loads_per_rec, int lanes_per_rec) {
    unsigned n_rec32 = (unsigned)(bytes / 32);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k_gather<MODE>, dim3(blocks), dim3(256), 0, 0, d, n_rec32, 4, dout);
    hipEventRecord(a);
    hipLaunchKernelGGL(k_gather<MODE>, dim3(blocks), dim3(256), 0, 0, d, n_rec32, iters, dout);
    hipEventRecord(b);
    CK(hipDeviceSynchronize());
    float ms; hipEventElapsedTime(&ms, a, b);
    double lane_loads = (double)blocks * 256 * iters * 4 * loads_per_rec;
    double recs = (double)blocks * 256 * iters * 4 / lanes_per_rec;
    printf("  %-44s %8.3f ms  %7.1f G lane-loads/s  %7.1f G records/s  %7.1f GB/s useful\n", name, ms, lane_loads / ms / 1e6, recs / ms / 1e6,
           lane_loads * 16 / ms / 1e6);
}

int main() {
    float* dout; CK(hipMalloc(&dout, 4));
    size_t sizes[] = {2ull << 20, 24ull << 20, 273ull << 20, 2048ull << 20};
    for (size_t bytes : sizes) {
        float4* d; CK(hipMalloc(&d, bytes)); CK(hipMemset(d, 0, bytes));
        for (int blocks : {1024, 2048}) {
            printf("working set %zu MiB, %d blocks of 256:\n", bytes >> 20, blocks);
            int iters = 64;
            run<0>("own 32-B record, 2 x 16 B per lane", d, bytes, blocks, iters, dout, 2, 1);
            run<1>("own record, 1 x 16 B per lane", d, bytes, blocks, iters, dout, 1, 1);
            run<2>("pair shares 32-B record (1 load/lane)", d, bytes, blocks, iters, dout, 1, 2);
            run<3>("quad shares 64-B block (1 load/lane)", d, bytes, blocks, iters, dout, 1, 4);
            run<4>("own 64-B block, 4 x 16 B per lane", d, bytes, blocks, iters, dout, 4, 1);
        }
        hipFree(d);
    }
    return 0;
}
